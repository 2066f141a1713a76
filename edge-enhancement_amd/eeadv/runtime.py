"""Process-wide switches of the host layer."""
import atexit
import os
import shutil
import tempfile

import torch

_CPU_PLUMBING = False


def allow_cpu_plumbing(on=True):
    """The HIP kernels are the product; CPU tensors are refused unless a driver opts in explicitly
    (`--no-cuda`, BASELINE config 1: MNIST ST on CPU).  The opt-in path is plain torch ops with the
    reference's own expressions - it exists for plumbing runs on a GPU-less host, is never taken for a
    ROCm tensor, and is never what the GPU tests or bench.py measure."""
    global _CPU_PLUMBING
    _CPU_PLUMBING = bool(on)


def cpu_plumbing_allowed():
    return _CPU_PLUMBING


def require_device(t, what):
    if t.is_cuda:
        return True
    if not _CPU_PLUMBING:
        raise RuntimeError(
            "%s received a %s tensor: the eeadv hot path runs on a ROCm device only. Move the data to the GPU, or call "
            "eeadv.runtime.allow_cpu_plumbing(True) for a torch-op plumbing run on the host (what --no-cuda does)."
            % (what, t.device))
    return False


def philox_ticket(device, n_elements):
    """(seed, offset) for a device-side random start, taken from - and advancing - torch's CUDA generator of
    `device`, so torch.manual_seed / set_seed govern it like they govern uniform_ in the reference."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + 4 * ((n_elements + 3) // 4))
    return seed & 0xFFFFFFFFFFFFFFFF, off // 4


_DRAW_STATE = {}
_DRAW_STALE = set()


def draw_state(device):
    """Device-resident {seed, offset, ticket, -} of the in-graph random draws (Add_Square): an int64[4] tensor whose offset the
    drawing kernel advances by itself (ee_square_draw_f32, ee_chain_fwd_f32: the last workgroup to finish, counted by the ticket),
    so that a replayed HIP graph never repeats its numbers.  Seeded from torch's CUDA generator of
    the device at first use (torch.manual_seed before the first forward governs it); reseed() makes the next use re-read the generator."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _DRAW_STATE.get(idx)
    if st is None or idx in _DRAW_STALE:
        if torch.cuda.is_current_stream_capturing():
            if st is None:
                raise RuntimeError("eeadv.runtime.draw_state: first use inside a graph capture (call it once eagerly, e.g. one eager forward)")
            return st  # re-seeding waits for the next eager use: the generator cannot be read during a capture
        seed, off = philox_ticket(torch.device("cuda", idx), 4)
        vals = torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed, off, 0, 0], dtype=torch.int64)
        if st is None:
            st = _DRAW_STATE[idx] = vals.to(torch.device("cuda", idx))
        else:
            st.copy_(vals)  # IN PLACE: captured graphs hold this tensor's address
        _DRAW_STALE.discard(idx)
    return st


def reseed():
    """The next draw_state() of every device re-reads torch's CUDA generator (into the SAME tensor: captured graphs keep drawing from it)."""
    _DRAW_STALE.update(_DRAW_STATE.keys())


def capture_mode():
    """capture_error_mode for torch.cuda.graph: with a process group alive its watchdog thread polls HIP events while we capture;
    under the default "global" mode a call from ANY thread that the runtime deems unsafe invalidates the capture, "thread_local"
    confines the check to the capturing thread (the launches that get captured are the same)."""
    import torch.distributed as dist
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


def non_default_switches():
    """Every EEADV_* environment switch that is set in this process, for the bench line's `config.switches`: a number measured with
    an A/B switch thrown says so.  EEADV_GRAPH is left out (bench.py sets it itself and reports `hip_graph`)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("EEADV_") and k != "EEADV_GRAPH"}


# ---- MIOpen's solver choices for the convolutions that stay on MIOpen --------------------------------------------------------------------
# torch.backends.cudnn.benchmark = True makes MIOpen time every applicable solver once per convolution shape and direction: 4.5 minutes of
# kernel compilation for ResNet-50's ~50 shapes on a fresh machine, for 7 % throughput on BASELINE config 5 (bench.py: 553 against 518 img/s).
# The outcome of that search is a 70 KB text file (MIOpen's "user find-db").  `miopen_db/` holds the one recorded on an MI355X with this
# image's MIOpen for the convolution shapes of the BASELINE configs (scripts/record_miopen_db.sh); with it in place the search is a lookup.
_MIOPEN_DB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miopen_db")


def use_shipped_miopen_db():
    """Point MIOpen's user find-db at a PRIVATE copy of `miopen_db/` (every process its own: MIOpen appends to the files, and eight ranks must
    not share them; removed again when the process exits).  Call before the first convolution.  A MIOPEN_USER_DB_PATH the caller has set
    wins; shapes the recorded db does not know are searched as usual.  Returns the directory, or None when nothing was done."""
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(_MIOPEN_DB):
        return None
    dst = tempfile.mkdtemp(prefix="eeadv_miopen_")
    atexit.register(shutil.rmtree, dst, ignore_errors=True)
    for name in os.listdir(_MIOPEN_DB):
        shutil.copy(os.path.join(_MIOPEN_DB, name), dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


def shipped_miopen_db_matched(db_dir, device=None):
    """Did MIOpen take the recorded find-db?  The shipped files are named after ONE MIOpen build and device
    (`<arch><CU count in hex>.HIP.<major>_<minor>_<patch>_<build>.ufdb.txt`); any other build or device opens files of ITS name in the
    same directory and finds them empty: it searches, or (without the search) takes its immediate-mode solvers - silently.  True when a
    shipped file carries this process's MIOpen version, architecture and CU count AND MIOpen created no file of another name in the
    private copy; False otherwise; None when there is no private copy."""
    if not db_dir or not os.path.isdir(db_dir):
        return None
    shipped = set(os.listdir(_MIOPEN_DB))
    try:
        v = int(torch.backends.cudnn.version())  # MIOpen: major * 1e6 + minor * 1e3 + patch
        props = torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device)
        head = "%s%x.HIP.%d_%d_%d_" % (props.gcnArchName.split(":")[0], props.multi_processor_count, v // 1000000, (v // 1000) % 1000, v % 1000)
    except Exception:  # noqa: BLE001 - no device / no MIOpen: nothing can have matched
        return False
    named = any(n.startswith(head) for n in shipped)
    foreign = any(n not in shipped and (n.endswith(".ufdb.txt") or n.endswith(".udb.txt")) for n in os.listdir(db_dir))
    return named and not foreign
