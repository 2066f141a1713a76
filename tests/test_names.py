"""A name check over the host layer (`convert_sync_batchnorm` was an undefined bare name on the world > 1 branch of the free-AT
script, which no test reached): every name a scope reads must be bound in that scope, an enclosing one, the module, or builtins.  The
standard library's symtable does the scoping; no code is imported or run."""
import builtins
import glob
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(
    glob.glob(os.path.join(ROOT, "edge-enhancement_amd", "**", "*.py"), recursive=True)
    + glob.glob(os.path.join(ROOT, "scripts", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py"))
    + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


def _unbound(table, module_names, path, out):
    for sym in table.get_symbols():
        if not sym.is_referenced():
            continue
        name = sym.get_name()
        if table.get_type() == "module":
            bound = sym.is_assigned() or sym.is_imported() or sym.is_namespace()
        else:
            # a free variable is bound by an enclosing function (symtable resolved it); a global one must exist at module level
            bound = sym.is_local() or sym.is_free() or sym.is_parameter() or sym.is_imported() or sym.is_namespace() or name in module_names
            if table.get_type() == "class" and name in ("__class__", "__module__", "__qualname__"):
                bound = True
        if not bound and not hasattr(builtins, name) and name not in ("__file__", "__name__", "__doc__", "__class__"):
            out.append("%s:%d: %s in %s %s" % (os.path.relpath(path, ROOT), table.get_lineno(), name, table.get_type(), table.get_name()))
    for child in table.get_children():
        _unbound(child, module_names, path, out)


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(p, ROOT) for p in FILES])
def test_every_name_is_bound(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    if "import *" in src:
        pytest.skip("star import: module-level names are not statically known")
    bad = []
    _unbound(top, module_names, path, bad)
    assert not bad, "\n".join(bad)
