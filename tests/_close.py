"""Element-wise float comparison used by the parity tests (north star: "box/logit floats within 1e-3 rel").

`rel_err` in earlier rounds was tensor-scale (max|d| / max|want|): a small probability could be off by far more than
1e-3 of itself and pass.  `assert_elementwise` bounds EVERY element: |got - want| <= rtol * |want| + atol, and on failure
reports the worst element (value, error, index) instead of a bare assertion."""
import numpy as np

RTOL = 1e-3
ATOL_PROB = 1e-6          # probabilities and boxes: quantities of order 1e-2 .. 1


def elementwise_report(got, want, rtol=RTOL, atol=ATOL_PROB):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got - want)
    slack = err - (rtol * np.abs(want) + atol)
    k = int(np.argmax(slack))
    idx = np.unravel_index(k, want.shape)
    rel = err / np.maximum(np.abs(want), 1e-300)
    return {"ok": bool(slack.flat[k] <= 0.0), "worst_index": tuple(int(i) for i in idx), "want": float(want.flat[k]), "got": float(got.flat[k]),
            "abs_err": float(err.flat[k]), "max_abs_err": float(err.max()), "max_rel_err_above_atol": float(rel[np.abs(want) > 100 * atol].max(initial=0.0))}


def assert_elementwise(got, want, what="", rtol=RTOL, atol=ATOL_PROB):
    r = elementwise_report(got, want, rtol, atol)
    assert r["ok"], f"{what}: element {r['worst_index']} want {r['want']:.9g} got {r['got']:.9g} |d| {r['abs_err']:.3g} > {rtol:g}*|want| + {atol:g}"
    return r


def logit_atol(want) -> float:
    """Logits cross zero, so a purely relative bound is undefined there; what a softmax / sigmoid sees is the ABSOLUTE error
    of a logit.  The absolute floor is 1e-3 of the tensor's RMS logit: an element passes when it is within 1e-3 of its own
    magnitude or within 1e-3 of the typical magnitude, whichever is larger."""
    want = np.asarray(want, np.float64)
    return RTOL * float(np.sqrt(np.mean(want * want)))


def assert_logits(got, want, what=""):
    return assert_elementwise(got, want, what, RTOL, logit_atol(want))


def _fp32_oracle_rel(want32, want64, atol) -> float:
    """Worst element-wise error of the CPU fp32 oracle against the fp64 one, as a relative tolerance (0 when it meets 1e-3)."""
    w32, w64 = np.asarray(want32, np.float64), np.asarray(want64, np.float64)
    need = (np.abs(w32 - w64) - atol) / np.maximum(np.abs(w64), 1e-300)
    return float(max(need.max(), 0.0))


def check_predictions(model_heads, y_pred, out, suffix="", probes=None, out32=None):
    """Probabilities, boxes (element-wise 1e-3 + 1e-6) and the three heads' pre-activation logits against the oracle's StepOut
    (`out`: the fp64 run).  model_heads = (category head, attribute head, box head) whose `last_logits` the step just produced.

    out32 (optional): the CPU fp32 oracle's StepOut for an ILL-CONDITIONED config (batch statistics over a few dozen samples):
    there the reference's own fp32 arithmetic misses 1e-3 element-wise against fp64, and the bound becomes "1e-3, or as close to
    the exact result as the fp32 reference itself gets" (rtol = max(1e-3, the fp32 oracle's worst element); both printed)."""
    reports = {}
    for name, got, attr in zip(("category", "attribute", "box"), y_pred, ("cat_preds", "attribute_preds", "box_preds")):
        want = getattr(out, attr).detach().numpy()
        rtol = RTOL
        if out32 is not None:
            rtol = max(RTOL, _fp32_oracle_rel(getattr(out32, attr).detach().numpy(), want, ATOL_PROB))
        reports[name] = assert_elementwise(got.detach().cpu().numpy(), want, name, rtol)
        reports[name]["rtol_used"] = rtol
    probes = out.probes if probes is None else probes
    for head, key in zip(model_heads, (f"CategoryPredictionHead{suffix}/logits", f"AttributePredictionHead{suffix}/logits", f"BoxPredictionHead{suffix}/logits")):
        want = probes[key].detach().numpy()
        atol, rtol = logit_atol(want), RTOL
        if out32 is not None:
            rtol = max(RTOL, _fp32_oracle_rel(out32.probes[key].detach().numpy(), want, atol))
        reports[key] = assert_elementwise(head.last_logits.detach().cpu().numpy(), want, key, rtol, atol)
        reports[key]["rtol_used"] = rtol
    return reports
