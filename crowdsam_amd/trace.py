"""roctx ranges for rocprofv3 (``rocprofv3 --marker-trace --kernel-trace -- python tools/test.py --profile ...``).

The reference has no tracing at all (two commented-out time.time() calls, crowdsam/model.py:413,424).
Ranges: ``generate`` > ``set_image`` (``sam_encoder`` / ``dinov2`` inside) / ``sample_prompts`` / ``eps_sweep``
(``decoder_batch`` inside) / ``gather`` / ``nms`` / ``small_regions`` / ``rle``, and ``lookahead`` for the work queued on the
side stream.  Off unless enable() is called: push / pop are two attribute tests then."""
import ctypes
import json

enabled = False
depth = 0
_lib = None


def enable():
    """Load the roctx library of the ROCm install (rocprofiler-sdk's, else the legacy one); False when neither is there."""
    global _lib, enabled
    if _lib is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                     "/opt/rocm/lib/libroctx64.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _lib is not None:
            _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            _lib.roctxMarkA.argtypes = [ctypes.c_char_p]
    enabled = _lib is not None
    return enabled


def disable():
    global enabled
    unwind(0)
    enabled = False


def push(name):
    global depth
    if enabled:
        _lib.roctxRangePushA(name.encode())
        depth += 1


def pop():
    global depth
    if enabled and depth > 0:
        _lib.roctxRangePop()
        depth -= 1


def mark(name):
    if enabled:
        _lib.roctxMarkA(name.encode())


def unwind(to_depth):
    """Close every range above ``to_depth`` (an exception or an early return inside a stage)."""
    while enabled and depth > to_depth:
        pop()


class range:            # noqa: A001 -- `with trace.range("sam_encoder"):`
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        push(self.name)

    def __exit__(self, *exc):
        pop()
        return False


def write_timings(path, rank, model, images, kept_masks, wall_s):
    """The machine-readable per-rank record of a --profile run (the reference only prints): per-stage
    milliseconds summed over this rank's images and per image, prompts and kept masks."""
    t = dict(getattr(model, "timings", {}) or {})
    n = max(1, int(images))
    rec = {"rank": int(rank), "images": int(images), "wall_s": float(wall_s), "images_per_s": int(images) / max(wall_s, 1e-9),
           "stage_ms_total": t, "stage_ms_per_image": {k: v / n for k, v in t.items()}, "kept_masks": int(kept_masks),
           "max_prompts_per_image": int(getattr(model, "max_prompts", 0)),
           "note": "stage times are device-synchronised: stages do not overlap each other or the look-ahead in a --profile run"}
    with open(path, "w") as f:
        json.dump(rec, f, indent=1)
    return rec
