"""Worst observed deviation per parity quantity of a test session (written to gpurun_out/parity_worst.json by conftest)."""
WORST = {}


def parity_log(name, value, bar):
    v = float(value)
    cur = WORST.get(name)
    if cur is None or v > cur["worst"]:
        WORST[name] = {"worst": v, "bar": float(bar)}
