"""Worst observed deviation per parity quantity of a test session (written to gpurun_out/parity_worst.json by conftest)."""
WORST = {}


def parity_log(name, value, bar):
    v = float(value)
    cur = WORST.get(name)
    if cur is None or v > cur["worst"]:
        WORST[name] = {"worst": v, "bar": float(bar)}


def hip_temporal_gates(model, x, f, xpad, fpad):
    """The ReLU gates one HIP forward of sais_amd.temporal.fullModel takes on (x, f), in the call order of
    oracle.sais_oracle.temporal_forward (per stream: the FFN gates of the 4 layers, then the aggregate ReLU; the head's ReLU
    last), for oracle.imposed_gates.  Dropout must be off (the saved FFN activation is the dropped one)."""
    import torch
    dev = (x if x is not None else f).device
    model._engine(dev)
    with torch.no_grad():
        _, _, _, saved = model._forward_kernels(None if x is None else x.detach(), None if f is None else f.detach(),
                                                None if x is None else model._mask(xpad, x, dev),
                                                None if f is None else model._mask(fpad, f, dev), save=True)
    gates = []
    for sname, zname in (("sr", "zr"), ("sf", "zf")):
        if saved[sname] is None:
            continue
        gates += [(layer["h"] > 0).cpu() for layer in saved[sname]["layers"]]
        gates.append((saved[zname] > 0).cpu())
    gates.append((saved["rep"] > 0).cpu())
    return gates
