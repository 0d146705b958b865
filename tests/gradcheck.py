"""TEST INFRASTRUCTURE.  Per-tensor gradient comparison used by the model-level GPU parity tests: relative L2 of every
parameter gradient against the oracle's autograd (gradient NORMS with a floor would pass a
mis-directed or small-tensor gradient).  Gradients that are zero by construction (a bias in front of an InstanceNorm /
LayerNorm, a softmax shift) have no meaningful relative error: they are recognised by the ORACLE's norm (below
`zero` x the largest gradient norm) and the device gradient is then held to an absolute bound instead."""


def compare_grads(named_device_grads, oracle_grads, tol, zero=1e-6, zero_abs=1e-4):
    """named_device_grads: iterable of (name, tensor or None); oracle_grads: dict name -> tensor.
    Returns (worst_rel, worst_name, failures)."""
    top = max(float(g.double().norm()) for g in oracle_grads.values())
    worst, worst_name, bad = 0.0, None, []
    for k, g in named_device_grads:
        want = oracle_grads[k].double()
        if g is None:
            bad.append((k, "no gradient"))
            continue
        got = g.detach().double().cpu()
        wn = float(want.norm())
        if wn < zero * top:                                   # zero by construction
            if float(got.norm()) > zero_abs * top:
                bad.append((k, f"|g| {float(got.norm()):.2e} where the oracle's is {wn:.2e}"))
            continue
        e = float((got - want).norm()) / wn
        if e > worst:
            worst, worst_name = e, k
        if not e < tol:
            bad.append((k, e))
    return worst, worst_name, bad
