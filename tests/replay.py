"""Step-by-step replay of a reference attack trajectory on the GPU (shared by the -m gpu tests).

The reference (run on the host when tests/golden/make_golden.py made the fixtures) recorded every iterate x_k it fed to
the model and the gradient g_k it got back.  A free-running attack on the GPU cannot be bit-identical to that run: the
classifier's convolutions are MIOpen's here and oneDNN's there, the two round differently, and a gradient entry whose
magnitude is inside that rounding noise may change sign - which moves the pixel by 2*alpha and, from the next step on,
every later iterate.  The replay separates the two effects.  For every recorded step k it feeds the REFERENCE's x_k to the
GPU model and requires

  (1) |g_gpu - g_ref| <= bound := tol * max|g_ref| + atol     (the model + loss-gradient kernel agree with the reference),
  (2) x_{k+1} from the HIP update kernel == the reference's x_{k+1} at every element with |g_ref| > bound
      (an element can only differ if sign(g_gpu) != sign(g_ref), which (1) allows only below that magnitude),
  (3) with the reference's own g_k the HIP update kernel reproduces x_{k+1} bit for bit everywhere.

So every pixel in which a free-running GPU attack may leave the reference trajectory is one whose reference gradient
was below `bound` at the step where it diverged.

That criterion assumes a gradient that depends continuously on rounding.  Networks with max-pooling do not have one: a
pooling window whose two largest entries differ by one ulp hands its gradient to a different pixel in another implementation,
and the input gradient changes by several per cent of its largest entry at once (Net_2 at step 16 of the recorded PGD-40 run:
2.6 %; oneDNN vs MIOpen).  For those models pass `model64`, the same network in float64: (1) becomes an error BUDGET,
    ||g_gpu - g_64|| <= 2 ||g_ref - g_64|| + 1e-6 ||g_64||     and     #[sign(g_gpu) != sign(g_64)] <= 2 #[sign(g_ref) != sign(g_64)] + n/1000
(the GPU path is no further from the exact gradient than twice the reference's own fp32 path is), and (2) is asserted wherever
the two fp32 implementations agree on the sign.  `switch_tol` > 0 additionally admits, per step, an error of that fraction of
||g_64||: the size of ONE pooling switch going the other way on the GPU while the reference happened to round like float64 (the
per-step statistics come back to the caller, which bounds how often that may happen).  `atol` is for the KL loops only: at their first step x_adv = x + 0.001 * randn,
so d KL / d logits = (softmax(z_adv) - softmax(z_nat)) / B is a difference of nearly equal fp32 numbers (entries ~1e-5) whose
rounding error (~1e-8, host libm vs device libm) is not small RELATIVE to the gradient it produces (largest entry ~1e-4).
"""
import numpy as np
import torch

from oracle import ee_oracle as O


def _loss64(spec, logits):
    """The scalar loss of an engine.LossSpec with torch ops in float64 (for the float64 reference gradient)."""
    import torch.nn.functional as F
    p = spec.payload
    if spec.kind == "ce_sum":
        return F.cross_entropy(logits, p, reduction="sum")
    if spec.kind == "ce_mean":
        return F.cross_entropy(logits, p)
    if spec.kind == "kl":
        return F.kl_div(F.log_softmax(logits, dim=1), F.softmax(p.double(), dim=-1), reduction="batchmean")
    if spec.kind == "softce":
        return -torch.sum(F.log_softmax(logits, dim=1) * p.double())
    raise ValueError(spec.kind)


def replay_trajectory(model, xs, gs, x0, spec, alpha, eps, direction=1, final=None, tol=1e-5, atol=0.0, lo=0.0, hi=1.0, dev="cuda:0",
                      model64=None, switch_tol=0.0):
    """Returns per-step dicts(err, undecided, flipped); raises AssertionError when (1)-(3) fail."""
    from eeadv import engine, ops
    x0_d = torch.from_numpy(np.ascontiguousarray(x0)).to(dev)
    stats = []
    for k in range(len(gs)):
        g_ref = gs[k]
        want = O.pgd_step(xs[k], g_ref, x0, alpha, eps, lo, hi, direction)
        nxt = xs[k + 1] if k + 1 < len(xs) else final
        if nxt is not None:
            assert np.array_equal(want, nxt), "oracle step %d is not the reference's" % k
        xk = torch.from_numpy(np.ascontiguousarray(xs[k])).to(dev)
        g = engine.input_gradient(model, xk.clone(), spec).detach().contiguous()
        g_np = g.cpu().numpy()
        scale = float(np.nanmax(np.abs(g_ref)))
        assert np.array_equal(np.isnan(g_np), np.isnan(g_ref)), "step %d: NaN pattern of the gradient differs" % k
        err = float(np.nanmax(np.abs(g_np - g_ref))) if g_ref.size else 0.0
        x_next = xk.clone()
        ops.pgd_step_(x_next, g, x0_d, alpha, eps, lo, hi, direction)
        got = x_next.cpu().numpy()
        if model64 is not None:
            x64 = xk.double().requires_grad_(True)
            with torch.enable_grad():
                (g64,) = torch.autograd.grad(_loss64(spec, model64(x64)), x64)
            g64 = g64.cpu().numpy()
            e_gpu, e_ref, n64 = np.linalg.norm(g_np - g64), np.linalg.norm(g_ref - g64), np.linalg.norm(g64)
            assert e_gpu <= max(2 * e_ref + 1e-6 * n64, switch_tol * n64), "step %d: GPU gradient %.3g from float64, the reference's %.3g (|g| %.3g)" % (k, e_gpu, e_ref, n64)
            f_gpu, f_ref = int((np.sign(g_np) != np.sign(g64)).sum()), int((np.sign(g_ref) != np.sign(g64)).sum())
            assert f_gpu <= 2 * f_ref + max(g64.size // 1000, int(switch_tol * g64.size)), "step %d: %d sign disagreements with float64, the reference has %d" % (k, f_gpu, f_ref)
            decided = np.sign(g_np) == np.sign(g_ref)
            extra = {"e_gpu": e_gpu / n64, "e_ref": e_ref / n64, "f_gpu": f_gpu, "f_ref": f_ref}
        else:
            bound = tol * scale + atol
            assert err <= bound, "step %d: gradient differs by %.3g (%.3g of its largest entry %.3g, allowed %.3g)" % (k, err, err / scale, scale, bound)
            decided = np.abs(np.nan_to_num(g_ref)) > bound
            extra = {}
        assert np.array_equal(got[decided], want[decided]), "step %d: an element with a decided gradient sign moved differently" % k
        x_ref = xk.clone()
        ops.pgd_step_(x_ref, torch.from_numpy(np.ascontiguousarray(g_ref)).to(dev), x0_d, alpha, eps, lo, hi, direction)
        assert np.array_equal(x_ref.cpu().numpy(), want), "step %d: update kernel is not bit-exact on the reference's gradient" % k
        stats.append(dict({"err": err / scale if scale else 0.0, "undecided": int((~decided).sum()), "flipped": int((got != want).sum()),
                           "n": int(want.size)}, **extra))
    return stats
