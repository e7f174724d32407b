"""castVARCHAR(float32 / float64) on the CPU: the device library's shortest-digit generator (host build: tests/host_devlib) against
numpy's Dragon4 "unique" digits, in the Java layout — every power of two and its neighbours, d * 10^k and neighbours, and a
strided sweep of ALL float32 bit patterns.        python tools/float_text_check.py [stride]"""
import ctypes as C, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tests", "host_devlib", "libhost_devlib.so"))


def layout(digits, k, neg):
    x, nd = k - 1, len(digits)
    if -3 <= x < 7:
        t = "0." + "0" * (-k) + digits if k <= 0 else digits + "0" * (k - nd) + ".0" if nd <= k else digits[:k] + "." + digits[k:]
    else:
        t = digits[0] + "." + (digits[1:] or "0") + "E" + str(x)
    return ("-" if neg else "") + t


def expect(v, is32):
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-Infinity" if v < 0 else "Infinity"
    neg = math.copysign(1.0, v) < 0
    if v == 0:
        return "-0.0" if neg else "0.0"
    s = np.format_float_scientific(np.float32(v) if is32 else np.float64(v), unique=True, trim="-")
    m, e = s.lstrip("-").split("e")
    return layout(m.replace(".", "").rstrip("0") or "0", int(e) + 1, neg)


def run(a, is32):
    n = len(a)
    out, ln = np.zeros(32 * n, np.uint8), np.zeros(n, np.int32)
    lib.host_real_text(int(is32), a.ctypes.data_as(C.c_void_p), C.c_long(n), C.c_long(100), out.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p))
    return [bytes(out[32 * i:32 * i + ln[i]]).decode() for i in range(n)]


def check(name, a, is32):
    bad = sum(1 for v, g in zip(a, run(a, is32)) if expect(float(v), is32) != g)
    print(f"{name}: {len(a)} values, {bad} mismatches", flush=True)


with np.errstate(all="ignore"):
    vals = []
    for e in range(-1074, 1024):
        v = np.ldexp(1.0, e)
        vals += [v, np.nextafter(v, 0), np.nextafter(v, np.inf)]
    for k in range(-323, 309):
        for d in (1, 2, 5, 9, 1.5, 9.999999999999999, 1.2345678901234567):
            v = float(d) * 10.0 ** k
            if np.isfinite(v) and v > 0:
                vals += [v, np.nextafter(v, 0), np.nextafter(v, np.inf)]
    check("float64: powers of two, d * 10^k, their neighbours", np.array(vals, np.float64), False)
    vals = []
    for e in range(-149, 128):
        v = np.float32(np.ldexp(1.0, e))
        vals += [v, np.nextafter(v, np.float32(0)), np.nextafter(v, np.float32(np.inf))]
    a = np.array(vals, np.float32)
    check("float32: powers of two and their neighbours", a[np.isfinite(a)], True)
    stride = int(sys.argv[1]) if len(sys.argv) > 1 else 997
    check(f"float32: every {stride}th bit pattern", np.arange(0, 2 ** 32, stride, dtype=np.uint64).astype(np.uint32).view(np.float32), True)
    rng = np.random.default_rng(1)
    check("float64: random bit patterns", rng.integers(0, 2 ** 64, 2_000_000, dtype=np.uint64).view(np.float64), False)
