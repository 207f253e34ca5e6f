"""Shared comparison of a finished `Learner.train` replay with a reference golden (tests/golden/train_*.npz written by
oracle/gen_golden.py): Adam moments and the WEIGHT DELTA (after - before; comparing `after` itself would hide a wrong
update behind the unchanged bulk of the weight) with relative tolerances plus a per-tensor absolute floor — an element
whose gradient is ~0 is the difference of large terms, its relative error is unbounded in ANY fp32 implementation —, the
per-step gradient norms, and the rule that >= 90 % of a tensor's delta elements agree within 2e-3 relative."""
import json
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def compare_post_train(learner, g, before, tag, *, m_rtol, v_rtol, d_rtol, gn_rtol, floor=2e-3, delta_outliers=0.0):
    """returns a report of the worst errors per quantity; asserts the tolerances of the module docstring.
    delta_outliers: fraction of a tensor's weight-delta elements that may miss the tolerance (32768-sample minibatches: a
    handful of the 1.6 M fc weights have a gradient within round-off of ZERO — dead ReLU inputs — and Adam's g / (|g| + eps)
    turns a 1e-9 difference there into a delta difference of a few percent of the largest delta; the float64 comparison of
    tests/test_gpu_parity_c2_c5.py shows the reference's own fp32 run has the same outliers)."""
    ac = learner.actor_critic
    sub = int(g["subsample"])
    after, m, v = ac.state_dict(), ac.flat_to_ref(learner.exp_avg), ac.flat_to_ref(learner.exp_avg_sq)
    rep = {"grad_norms": [float(x) for x in learner._grad_norms], "ref_grad_norms": [float(x) for x in g["grad_norms"]]}
    worst, checks, fracs = {}, [], []
    for name in g["param_names"]:
        name = str(name)
        gm, gv, gd = g["m_" + name], g["v_" + name], g["delta_" + name]
        am = m[name].reshape(-1)[::sub].double().numpy()
        av = v[name].reshape(-1)[::sub].double().numpy()
        ad = (after[name].double() - before[name].double()).reshape(-1)[::sub].numpy()
        # absolute floors: a per-tensor scale (an element whose gradient is ~0 is the difference of large terms)
        m_atol = floor * float(np.abs(gm).max()) + 1e-12
        v_atol = floor * float(np.abs(gv).max()) + 1e-20
        d_atol = 25 * floor * float(np.abs(gd).max()) + 1e-12
        worst[name] = dict(
            m=float((np.abs(am - gm) / (np.abs(gm) * m_rtol + m_atol)).max()),
            v=float((np.abs(av - gv) / (np.abs(gv) * v_rtol + v_atol)).max()),
            d=float((np.abs(ad - gd) / (np.abs(gd) * d_rtol + d_atol)).max()),
            m_maxmax=float(np.abs(am - gm).max() / np.abs(gm).max()),
            v_maxmax=float(np.abs(av - gv).max() / np.abs(gv).max()),
            d_maxmax=float(np.abs(ad - gd).max() / np.abs(gd).max()),
            d_frac_within_2e3=float((np.abs(ad - gd) <= 2e-3 * np.abs(gd) + 1e-10).mean()))
        if len(gd) >= 64:
            fracs.append((name, worst[name]["d_frac_within_2e3"]))
        checks += [(am, gm, m_rtol, m_atol, f"exp_avg {name}"), (av, gv, v_rtol, v_atol, f"exp_avg_sq {name}"),
                   (ad, gd, d_rtol, d_atol, f"weight delta {name}")]
    rep["worst_error_over_tolerance"] = worst
    try:
        os.makedirs(OUT, exist_ok=True)
        json.dump(rep, open(os.path.join(OUT, f"parity_{tag}.json"), "w"), indent=1)
    except OSError:
        pass
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=gn_rtol)
    assert all(fr >= 0.9 for _, fr in fracs), fracs
    for a, b, rtol, atol, msg in checks:
        if delta_outliers > 0 and msg.startswith("weight delta"):
            bad = np.abs(a - b) > atol + rtol * np.abs(b)
            assert bad.mean() <= delta_outliers, (msg, float(bad.mean()), int(bad.sum()))
            continue
        np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)
    return rep
