"""Measured parity figures of a GPU test run, kept next to the pass / fail: every test that holds a bar over a measured
quantity (worst relative gradient error, near-tie flips, ...) calls record(key, value=..., bar=...) and the run leaves
gpurun_out/parity_measurements.json (merged over the processes of a run).  The committed copy of a round is
profiles/rNN_gradient_errors.json; the bars in the tests are set from it (4x the measurement)."""
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.environ.get("MCQ_RECORD_DIR", os.path.join(_ROOT, "gpurun_out"))


def record(key: str, **vals) -> None:
    try:
        os.makedirs(_DIR, exist_ok=True)
        path = os.path.join(_DIR, "parity_measurements.json")
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[key] = {k: (float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v) for k, v in vals.items()}
        with open(path + ".tmp", "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
        os.replace(path + ".tmp", path)
    except OSError:
        pass                                    # (a read-only checkout: the measurement is still in the assertion message)
