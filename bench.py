#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric on BASELINE.json's config — and, in the same line ("workloads"), the
other BASELINE configs (C1, C3, C4, C5) and the fused filter -> project (K2F), each timed, verified and priced
against the roofline by the same run (round 6).

  metric   million rows/s (+ achieved HBM GB/s in `roofline`) of the 10-expression float64
           Projector with 10 % nulls per column (config C2), 2^28 rows PER GPU.
  step     one Projector::Evaluate over the HBM-resident batch = one launch of the fused
           projection kernel (inputs/outputs resident and pre-touched; no PCIe in the
           timed region).
  N GPUs   one process per GPU (torch.distributed / RCCL used only for the timing barrier);
           rows are range-sharded, every rank evaluates its own 2^28-row shard, no
           data-path collective -> weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      algorithmic bytes (113.75 B/row: every input read once, every output
                written once, bitmaps at 1 bit/row) / mean kernel time measured with HIP
                events on the launch stream, against the 8 TB/s HBM3E peak
  cpu_baseline  the CPU restatement (oracle/, "port") timed on this host's cores on a
                bounded prefix of the same workload.

One workload alone (for profiling):  --workload c3 | c4 | c5 | k2f | c1      --no-extras: the headline without "workloads"
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "k2f"])
    p.add_argument("--rows", type=int, default=0, help="rows per GPU (default: BASELINE size)")
    p.add_argument("--cpu-rows", type=int, default=1 << 26, help="rows of the cpu_baseline sample")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--inproc", action="store_true",
                   help="ONE process, --gpus host threads, each on its own device context (round 3: "
                        "in-process multi-device; virtual contexts when the box has fewer GPUs). c2 only")
    p.add_argument("--data", default="pcg64", choices=["pcg64", "philox"],
                   help="c2 inputs: pcg64 = BASELINE.md §4's frozen numpy PCG64 streams (value seeds 42-45, mask "
                        "seeds 142-145; generated on the host cores, uploaded before the timed region); "
                        "philox = same distributions from torch's device generator (faster to set up)")
    p.add_argument("--pool-candidates", type=int, default=8,
                   help="projection workloads: the output columns come from the library's device pool (gdv_device_pool_reserve_set), "
                        "which allocates up to this many candidate placements of the whole set, probes each with a write sweep and "
                        "keeps the fastest (round 6: the placement search is the product's, not the bench's); 0 = plain allocations")
    p.add_argument("--placements", type=int, default=1,
                   help="projection workloads: allocate the batch's columns and outputs this many times, time 3 steps on "
                        "each placement and keep the fastest (profiles/r05_box_states.txt: where the driver puts the "
                        "buffers moves a C2 step between 4.9 and 7.0 ms, and stays with the buffers); 1 = off (the default since round 6: "
                        "the library's pool does the search, --pool-candidates)")
    p.add_argument("--no-verify", action="store_true",
                   help="skip the post-loop check of the outputs the timed loop produced")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: --rows per GPU (default); strong: ONE logical batch of --rows rows (default: BASELINE's size "
                        "for the workload: C2 2^28, C3 10^9, C4 6*10^9) row-sharded across the ranks by gandiva_amd.shard")
    p.add_argument("--no-extras", action="store_true",
                   help="default invocation only: leave out the 'workloads' object (the other BASELINE configs + K2F)")
    p.add_argument("--extras", default="c1,c3,k2f,c5,c4", help="sub-workloads of the default line, in this order")
    p.add_argument("--sub-steps", type=int, default=30)
    p.add_argument("--sub-warmup", type=int, default=3)
    p.add_argument("--sub-placements", type=int, default=3, help="placement trials of the projection sub-workloads (c1, c4)")
    p.add_argument("--sub-cpu-seconds", type=float, default=3.0)
    p.add_argument("--c5-sync", action="store_true",
                   help="c5: time the synchronous entry point (reads the byte totals back every step) instead of gdv_projector_evaluate_async")
    p.add_argument("--data-c5", default="philox", choices=["philox", "pcg64"],
                   help="c5 inputs: philox = BASELINE.md §4's distributions from torch's device generator (seconds); "
                        "pcg64 = its frozen numpy stream (generated on one host core: ~1 min for 10^8 rows)")
    return p.parse_args()


def effective_cores():
    """Cores this process may actually use: the scheduler affinity capped by the cgroup CPU
    quota (a 256-CPU host with cpu.max = 16 CPUs runs 16 threads at full speed and 256
    threads at a fraction of it — measured, tools/cpu_scaling.py)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, -(-q // per)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def host_description():
    """CPU model, logical CPUs, the cores this process may use and the load average — the facts a
    reader needs to compare two `cpu_baseline` figures taken on different boxes (round 3: 338 vs 822 M
    rows/s, same code, '16 cores' both times, and nothing in the line to tell the hosts apart)."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        load = [float(x) for x in open("/proc/loadavg").read().split()[:3]]
    except (OSError, ValueError):
        load = None
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cores": effective_cores(), "loadavg": load}


def timed_passes(fn, rows, budget_s, max_reps):
    """rows/s of repeated calls of fn() for about budget_s seconds (at least one call)."""
    t0 = time.perf_counter()
    reps = 0
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= max_reps:
            return rows * reps / el, reps, el


def cpu_baseline_c2(rows):
    """Oracle ("port"), expression-at-a-time like the reference: one thread, then all usable cores
    three times (min / median / max reported; `value` = the median)."""
    from gandiva_amd import workloads as W
    from oracle import oracle
    host = host_description()
    cores = host["usable_cores"]
    batch = W.c2_batch(rows)
    exprs = W.c2_expressions()
    outs = oracle.alloc_outputs(exprs, rows)          # pre-touched, reused by every pass
    oracle.project(exprs, batch, threads=cores, out=outs)  # warm up
    one, reps1, el1 = timed_passes(lambda: oracle.project(exprs, batch, threads=1, out=outs), rows, 2.5, 50)
    runs, total_reps, total_el = [], 0, 0.0
    for _ in range(3):
        r, reps, el = timed_passes(lambda: oracle.project(exprs, batch, threads=cores, out=outs), rows, 3.0, 100)
        runs.append(r / 1e6)
        total_reps += reps
        total_el += el
    runs.sort()
    host["loadavg_after"] = host_description()["loadavg"]
    out = {
        "value": round(runs[1], 2),
        "unit": "million rows/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{total_reps} passes x {rows} rows, {cores} thr, {total_el:.0f}s (+{el1:.0f}s 1 thr)",
        "repeats_all_cores": {"min": round(runs[0], 2), "median": round(runs[1], 2), "max": round(runs[2], 2)},
        "one_thread": round(one / 1e6, 2),
        "per_thread_at_all_cores": round(runs[1] / cores, 2),
        "host": host,
        "what": "oracle/gdv_oracle.c -O3 -march=native (the builder's CPU restatement, NOT Gandiva's LLVM JIT), "
                "the first rows of the C2 generator, same ten expressions, 10% nulls",
    }
    try:
        out["second_engine"] = pyarrow_compute_c2(batch.slice(0, min(rows, 1 << 24)))
    except Exception as e:  # indicative only
        out["second_engine"] = {"engine": "pyarrow.compute", "value": None, "note": f"failed: {e}"}
    return out


def pyarrow_compute_c2(batch):
    """An independent CPU engine on the same data: pyarrow.compute's
    vectorised kernels, one call per operator, common sub-expressions shared by hand."""
    import pyarrow as pa
    import pyarrow.compute as pc
    a, b, c, d = batch.columns

    def once():
        e0, e1, e2, e3, e4 = pc.add(a, b), pc.subtract(a, b), pc.multiply(a, b), pc.add(c, d), pc.multiply(c, d)
        return [e0, e1, e2, e3, e4, pc.multiply(e0, c), pc.multiply(e1, d), pc.add(e2, e4),
                pc.multiply(e0, pc.subtract(c, d)), pc.multiply(pc.multiply(e2, c), d)]
    once()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.5 and reps < 50:
        once()
        reps += 1
    el = time.perf_counter() - t0
    return {"engine": f"pyarrow.compute {pa.__version__}, 13 kernel calls per pass, 1 thread",
            "value": round(batch.num_rows * reps / el / 1e6, 2), "unit": "million rows/s",
            "sample": f"{reps} passes over {batch.num_rows} rows, {el:.1f} s"}


def load_traffic(tag, running_kernel):
    """HBM bytes per launch from the committed PMC passes (tools/summarize_prof.py pmc) — quoted
    ONLY when they were taken on the very kernel that is running now (same source hash);
    returns (bytes or None, where the figure comes from)."""
    path = os.path.join(ROOT, "profiles", f"pmc_{tag}.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, "no PMC pass committed for this workload"
    if running_kernel is None or d.get("kernel") != running_kernel:
        return None, (f"profiles/pmc_{tag}.json was taken on {d.get('kernel')}, this run is "
                      f"{running_kernel}: not quoted")
    return d.get("hbm_bytes_per_launch"), f"profiles/pmc_{tag}.json@{d.get('kernel')}"


class BoxSampler:
    """Clocks, power and temperature of the GPU while the timed loop runs (round 3: the same binary
    measured 4.84 .. 6.20 ms on C2 across boxes of the pool — a reader must be able to tell a slow
    box from a slow kernel).  Reads the amdgpu sysfs / hwmon files every few milliseconds from a
    thread (microseconds per read; rocm-smi is a Python program that takes longer than the whole
    timed region).  Everything is best effort: a missing file is a missing key."""

    def __init__(self, pci_bus_id=None):
        import glob
        self.dev = None
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                if open(os.path.join(card, "vendor")).read().strip() != "0x1002":
                    continue
            except OSError:
                continue
            real = os.path.realpath(card)
            if pci_bus_id and pci_bus_id.lower() not in real.lower():
                continue
            self.dev = card
            break
        self.hwmon = None
        if self.dev:
            hw = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*")))
            self.hwmon = hw[0] if hw else None
        self.samples = []
        self._stop = False
        self._thread = None
        # hwmon reads go to the SMU: keep them rare (GDV_BENCH_TELEMETRY_MS, default 25 ms)
        self.interval = max(0.002, float(os.environ.get("GDV_BENCH_TELEMETRY_MS", "25")) / 1e3)

    @staticmethod
    def _read(path, scale):
        try:
            return float(open(path).read().split()[0]) * scale
        except (OSError, ValueError, IndexError):
            return None

    @staticmethod
    def _active_mhz(path):
        """pp_dpm_sclk / pp_dpm_mclk: the line marked '*' is the current level."""
        try:
            for line in open(path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().split("M")[0])
        except (OSError, ValueError, IndexError):
            pass
        return None

    def sample(self):
        s = {}
        if self.hwmon:
            s["sclk_mhz"] = self._read(os.path.join(self.hwmon, "freq1_input"), 1e-6)
            s["mclk_mhz"] = self._read(os.path.join(self.hwmon, "freq2_input"), 1e-6)
            p = self._read(os.path.join(self.hwmon, "power1_average"), 1e-6)
            s["power_w"] = p if p is not None else self._read(os.path.join(self.hwmon, "power1_input"), 1e-6)
            s["temp_edge_c"] = self._read(os.path.join(self.hwmon, "temp1_input"), 1e-3)
            s["temp_junction_c"] = self._read(os.path.join(self.hwmon, "temp2_input"), 1e-3)
            s["temp_mem_c"] = self._read(os.path.join(self.hwmon, "temp3_input"), 1e-3)
        if self.dev:
            if s.get("sclk_mhz") is None:
                s["sclk_mhz"] = self._active_mhz(os.path.join(self.dev, "pp_dpm_sclk"))
            if s.get("mclk_mhz") is None:
                s["mclk_mhz"] = self._active_mhz(os.path.join(self.dev, "pp_dpm_mclk"))
            s["gpu_busy_pct"] = self._read(os.path.join(self.dev, "gpu_busy_percent"), 1.0)
        return {k: v for k, v in s.items() if v is not None}

    def start(self):
        import threading

        def loop():
            while not self._stop:
                x = self.sample()
                if x:
                    self.samples.append(x)
                time.sleep(self.interval)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread:
            self._thread.join(timeout=1.0)

    def summary(self):
        out = {"samples": len(self.samples), "source": "amdgpu sysfs (hwmon)" if self.hwmon else
               ("amdgpu sysfs" if self.dev else "unavailable")}
        keys = sorted({k for s in self.samples for k in s})
        for k in keys:
            vals = [s[k] for s in self.samples if k in s]
            out[k] = {"min": round(min(vals), 1), "mean": round(sum(vals) / len(vals), 1), "max": round(max(vals), 1)}
        return out


VERIFY_WINDOW = 100_000


def verify_outputs(workload, rows, dbatch, result):
    """Look at what the timed loop left in HBM (the verdict of round 3: a bench that never reads the
    outputs it timed cannot tell a fast kernel from a broken one).  Two 10^5-row windows — the head of
    the batch and its middle — of EVERY output, recomputed on the device by an independent engine (torch
    elementwise kernels: IEEE float64 / wrapping integer arithmetic, nonzero, cumsum) and compared bit
    for bit.  The oracle-backed checks at full size live in tests/test_full_size.py; this one runs in
    the bench process itself, after the timed region.  Returns the `verified` object of the JSON line."""
    import torch
    from gandiva_amd import workloads as W
    windows = sorted({0, (rows // 2) & ~63})
    checked = 0

    def bits_equal(got, want, m):
        nb = m // 8
        if not torch.equal(got[:nb], want[:nb]):
            return False
        if m % 8:
            mask = (1 << (m % 8)) - 1
            return (int(got[nb]) & mask) == (int(want[nb]) & mask)
        return True

    for lo in windows:
        m = min(VERIFY_WINDOW, rows - lo)
        if m <= 0:
            continue
        if workload == "c2":
            vals, valid = W.c2_expected_window(dbatch, lo, m)
            for e, (o, v, vb) in enumerate(zip(result, vals, valid)):
                got = o.data.view(torch.int64)[lo:lo + m]
                if not torch.equal(got, v.view(torch.int64)):
                    return {"ok": False, "what": f"C2 output e{e} differs from torch float64 in rows [{lo}, {lo + m})"}
                if not bits_equal(o.validity[lo // 8:], vb, m):
                    return {"ok": False, "what": f"C2 validity of e{e} differs from the AND of its inputs' in rows [{lo}, {lo + m})"}
        elif workload == "c1":
            a, b, c = (col.data.view(torch.int32)[lo:lo + m] for col in dbatch.columns)
            if not torch.equal(result[0].data.view(torch.int32)[lo:lo + m], (a + b) * c):
                return {"ok": False, "what": f"C1 output differs from torch int32 in rows [{lo}, {lo + m})"}
        elif workload == "c4":
            ep, disc, tax = (dbatch.columns[k].data.view(torch.int64).view(-1, 2)[lo:lo + m] for k in range(3))
            ship = dbatch.columns[3].data.view(torch.int32)[lo:lo + m]
            dp = result[0].data.view(torch.int64)[:2 * rows].view(-1, 2)[lo:lo + m]
            ch = result[1].data.view(torch.int64)[:2 * rows].view(-1, 2)[lo:lo + m]
            days = result[2].data.view(torch.int32)[lo:lo + m]
            want = ep[:, 0] * (100 - disc[:, 0])      # fits 64 bits for this data: high words must be zero
            ok = (torch.equal(dp[:, 0], want) and not bool(dp[:, 1].any())
                  and torch.equal(ch[:, 0], want * (100 + tax[:, 0])) and not bool(ch[:, 1].any())
                  and torch.equal(days, W.C4_DATE_1998_12_01 - ship))
            if not ok:
                return {"ok": False, "what": f"C4 outputs differ from torch int64 arithmetic in rows [{lo}, {lo + m})"}
        elif workload == "c3":
            a, b = (col.data.view(torch.int64)[lo:lo + m] for col in dbatch.columns)
            want = torch.nonzero((a > W.C3_K1) & (b < W.C3_K2)).view(-1) + lo
            idx = result.indices[:result.num_slots].view(torch.int32).to(torch.int64) & 0xffffffff
            first = int(torch.searchsorted(idx, torch.tensor([lo], device=idx.device, dtype=torch.int64)))
            if not torch.equal(idx[first:first + want.numel()], want) or (
                    first + want.numel() < idx.numel() and int(idx[first + want.numel()]) < lo + m):
                return {"ok": False, "what": f"C3 selection vector differs from torch.nonzero in rows [{lo}, {lo + m})"}
        elif workload == "k2f":
            outs, sel = result
            a, b = (col.data.view(torch.int64)[lo:lo + m] for col in dbatch.columns)
            keep = (a > W.C3_K1) & (b < W.C3_K2)
            want = torch.nonzero(keep).view(-1) + lo
            count = sel.num_slots
            idx = sel.indices[:count].view(torch.int32).to(torch.int64) & 0xffffffff
            first = int(torch.searchsorted(idx, torch.tensor([lo], device=idx.device, dtype=torch.int64)))
            vals = outs[0].data.view(torch.int64)[first:first + want.numel()]
            nb = want.numel() // 8
            vbits = outs[0].validity[(first + 7) // 8:(first + 7) // 8 + max(nb - 2, 0)]
            if (not torch.equal(idx[first:first + want.numel()], want)
                    or (first + want.numel() < idx.numel() and int(idx[first + want.numel()]) < lo + m)
                    or not torch.equal(vals, (a + b)[keep]) or not bool((vbits == 255).all())):
                return {"ok": False, "what": f"K2F selection vector / compacted a+b differ from torch in rows [{lo}, {lo + m})"}
        elif workload == "c5":
            like, sub, up = result
            off = dbatch.columns[0].offsets.view(torch.int32)[lo:lo + m + 1].to(torch.int64)
            dat = dbatch.columns[0].data[int(off[0]):int(off[-1])]
            up_off = up.offsets.view(torch.int32)[lo:lo + m + 1].to(torch.int64)
            lower = (dat >= 97) & (dat <= 122)
            lens = off[1:] - off[:-1]
            sub_off = sub.offsets.view(torch.int32)[lo:lo + m + 1].to(torch.int64)
            sub_len = torch.clamp(lens - 1, min=0, max=5)
            # substr bytes: byte j of row r = input byte 1 + j of row r
            rep = torch.repeat_interleave(torch.arange(m, device=dat.device), sub_len)
            within = torch.arange(int(sub_len.sum()), device=dat.device) - torch.repeat_interleave(
                torch.cumsum(sub_len, 0) - sub_len, sub_len)
            want_sub = dbatch.columns[0].data[off[:-1][rep] + 1 + within]
            # like '%spark%': five-byte matches that end inside their row
            hit = torch.ones(max(dat.numel() - 4, 0), dtype=torch.bool, device=dat.device)
            for k, chv in enumerate(b"spark"):
                hit &= dat[k:dat.numel() - 4 + k] == chv
            pos = torch.nonzero(hit).view(-1) + int(off[0])
            r = torch.searchsorted(off, pos, right=True) - 1
            inside = pos + 5 <= off[r + 1]
            want_like = torch.zeros(m, dtype=torch.bool, device=dat.device)
            want_like[r[inside]] = True
            nb = (m + 7) // 8
            gb = like.data[lo // 8: lo // 8 + nb]
            got_like = ((gb.view(-1, 1) >> torch.arange(8, device=dat.device, dtype=torch.uint8)) & 1).view(-1)[:m].bool()
            ok = (torch.equal(up_off, off)
                  and torch.equal(up.data[int(off[0]):int(off[-1])], torch.where(lower, dat - 32, dat))
                  and torch.equal(sub_off[1:] - sub_off[:-1], sub_len)
                  and torch.equal(sub.data[int(sub_off[0]):int(sub_off[-1])], want_sub)
                  and torch.equal(got_like, want_like))
            if not ok:
                return {"ok": False, "what": f"C5 outputs differ from the torch restatement in rows [{lo}, {lo + m})"}
        checked += m
    return {"ok": True, "rows_checked": checked, "windows": [[lo, min(VERIFY_WINDOW, rows - lo)] for lo in windows],
            "against": "torch elementwise kernels on the device (independent of the library and of oracle/), "
                       "every output, values and validity, bit for bit"}



def hbm_ceilings(bytes_per_buffer=4 << 30):
    """Read-only / write-only / copy rate of plain streaming kernels on THIS box, in THIS process
    (gdv_device_hbm_ceilings: ~0.1 s)."""
    import ctypes as C
    from gandiva_amd import _capi
    r, w, c = C.c_double(), C.c_double(), C.c_double()
    if _capi.lib().gdv_device_hbm_ceilings(bytes_per_buffer, C.byref(r), C.byref(w), C.byref(c)) != 0:
        return None
    return {"read": round(r.value, 1), "write": round(w.value, 1), "copy": round(c.value, 1), "unit": "GB/s",
            "bytes_per_buffer": bytes_per_buffer}


# the workload's traffic as (input streams, output streams) of 8-byte elements, for the shape-matched ceiling
CEILING_SHAPE = {"c1": (3, 1), "c2": (4, 10), "c3": (2, 0), "c4": (7, 5), "c5": (2, 3), "k2f": (2, 0)}


def stream_ceiling(num_read, num_write, bytes_per_stream=1 << 30):
    """Best rate of the projection kernel's skeleton WITHOUT its arithmetic on this traffic shape
    (gdv_device_stream_ceiling: grid sizes x {plain, non-temporal} swept, ~0.3 s)."""
    import ctypes as C
    from gandiva_amd import _capi
    g, wg, nt = C.c_double(), C.c_int(), C.c_int()
    if _capi.lib().gdv_device_stream_ceiling(bytes_per_stream, num_read, num_write, C.byref(g), C.byref(wg), C.byref(nt)) != 0:
        return None
    return {"reads": num_read, "writes": num_write, "GB/s": round(g.value, 1), "workgroups_per_cu": wg.value % 100,
            "subtiles_per_wave": 16 if wg.value > 100 else 4, "nontemporal": bool(nt.value), "bytes_per_stream": bytes_per_stream}


def stream_ceiling_on(read_tensors, write_tensors, elems):
    """The same sweep on the buffers the timed loop used (the written ones are overwritten: call it
    after the verification)."""
    import ctypes as C
    from gandiva_amd import _capi
    ptrs = (C.c_void_p * (len(read_tensors) + len(write_tensors)))(*[t.data_ptr() for t in list(read_tensors) + list(write_tensors)])
    g, wg, nt = C.c_double(), C.c_int(), C.c_int()
    if _capi.lib().gdv_device_stream_ceiling_on(ptrs, len(read_tensors), len(write_tensors), elems, C.byref(g), C.byref(wg),
                                                C.byref(nt)) != 0:
        return None
    return {"reads": len(read_tensors), "writes": len(write_tensors), "GB/s": round(g.value, 1), "workgroups_per_cu": wg.value % 100,
            "subtiles_per_wave": 16 if wg.value > 100 else 4, "nontemporal": bool(nt.value), "bytes_per_stream": elems * 8,
            "buffers": "the timed loop's own"}


def kernel_name_of(obj):
    import re
    m = re.search(r"gdv_k_[0-9a-f]{16}", obj.llvm_ir)
    return m.group(0) if m else None


def main_inproc(args):
    """One process drives N device contexts with one host thread each: every thread
    selects its device (gandiva.set_device), generates its own 2^28-row shard there and evaluates the
    SAME Projector handle — the handle is loaded onto a context the first time it runs there.  Same
    barrier / max-over-threads timing as the multi-process launch; no collective, weak scaling.  With
    fewer physical GPUs than threads the contexts are virtual and share the GPUs (the value is then
    a correctness / plumbing figure, not a scaling one)."""
    import threading
    import torch
    import gandiva_amd as gandiva
    from gandiva_amd import workloads as W
    n = args.gpus
    rows = args.rows or (1 << 28)
    physical = gandiva.physical_device_count()
    if physical < 1:
        raise SystemExit("bench.py needs a HIP device")
    if n > physical:
        gandiva.set_virtual_devices(n)
        if not args.rows:
            rows = max((1 << 28) // n, 1 << 20)   # the shards share one GPU's HBM
    proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
    bar = threading.Barrier(n)
    elapsed = [0.0] * n
    errors = []

    def work(r):
        try:
            gandiva.set_device(r)
            dev = f"cuda:{r % physical}"
            with torch.cuda.device(r % physical):
                dbatch = W.c2_device_batch(rows, device=dev, seed_offset=1000 * r)
                outs = proj.evaluate_device(dbatch)
                for _ in range(args.warmup):
                    proj.evaluate_device(dbatch, outputs=outs, sync=False)
                torch.cuda.synchronize()
                bar.wait()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    proj.evaluate_device(dbatch, outputs=outs, sync=False)
                torch.cuda.synchronize()
                bar.wait()
                elapsed[r] = time.perf_counter() - t0
        except Exception as e:
            errors.append((r, repr(e)))
            try:
                bar.abort()
            except Exception:
                pass
    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit(f"in-process bench failed: {errors}")
    el = max(elapsed)
    total_rows = rows * n
    print(json.dumps({
        "metric": "million rows/sec, 10-expr float64 Projector (10% nulls)", "value": round(total_rows * args.steps / el / 1e6, 1),
        "unit": "million rows/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2: 10 float64 arithmetic expressions over 4 columns, 10% nulls per column",
                   "rows_per_gpu": rows, "total_rows": total_rows,
                   "sharding": f"row-range x{n}, ONE process, one host thread + device context per shard, no collective",
                   "physical_gpus": physical, "virtual_contexts": n > physical,
                   "residency": "inputs and outputs in HBM (zero-copy C-ABI path)"}}), flush=True)


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, exactly as the
    driver's torch.distributed.run line does (one process per GPU, 127.0.0.1 rendezvous).  Refuses to run —
    instead of printing a mislabelled one-rank line — when the node has fewer than N GPUs (unless the
    backend is gloo: the ranks then share cuda:0, a control-flow test, and the line says so)."""
    import socket
    import subprocess
    import torch
    backend = os.environ.get("GDV_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if have < 1:
        raise SystemExit("bench.py needs a HIP device: gandiva_amd has no CPU evaluation path")
    if backend == "nccl" and have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); refusing to run fewer "
                         f"ranks than asked for (GDV_BENCH_BACKEND=gloo shares one GPU between the ranks)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------
# workloads: set-up, placement search, timed loop, one (sub-)line each

WORKLOAD_TEXT = {
    "c1": ("million rows/sec, (a+b)*c int32 Projector", "int32", "C1 shape at scale: (a+b)*c over int32, no nulls"),
    "c2": ("million rows/sec, 10-expr float64 Projector (10% nulls)", "f64",
           "C2: 10 float64 arithmetic expressions over 4 columns, 10% nulls per column"),
    "c3": ("million rows/sec, Filter a>k1 AND b<k2 -> SelectionVector (int64)", "int64",
           "C3: filter a>499 AND b<250, int64 U[0,1000), uint32 selection vector"),
    "c4": ("million rows/sec, TPC-H Q1 projections (decimal128 + datediff)", "decimal128",
           "C4: ep*(1-disc), ep*(1-disc)*(1+tax) decimal128(15,2) inputs, datediff(1998-12-01, shipdate date32)"),
    "c5": ("million rows/sec, utf8 like/substr/upper", "u8",
           "C5: like '%spark%', substr(s,2,5), upper(s) over utf8 lengths U[4,20]"),
    "k2f": ("million rows/sec, fused Filter -> Projector (a>k1 AND b<k2; a+b; uint32 selection vector)", "int64",
            "K2F: C3's condition, a + b projected for the selected rows and the uint32 selection vector, ONE kernel "
            "(SURVEY §8 f3: the selection-vector consumer)"),
}
DEFAULT_ROWS = {"c1": 1 << 28, "c2": 1 << 28, "c3": 1_000_000_000, "c4": 750_000_000, "c5": 100_000_000,
                "k2f": 1_000_000_000}
# BASELINE.json's logical sizes for --scaling strong: ONE batch of this many rows row-sharded over the ranks
STRONG_ROWS = {"c1": 1 << 28, "c2": 1 << 28, "c3": 1_000_000_000, "c4": 6_000_000_000, "c5": 100_000_000,
               "k2f": 1_000_000_000}
AOT_KERNELS = {"c3": ["ScanReduce", "ScanSpine", "ScanApply", "EmitIndices"], "c5": ["the AOT segmented offsets scan"],
               "k2f": ["PublishCount"]}


def tensors_of(dbatch, outs):
    ts = []
    for c in list(dbatch.columns) + list(outs or []):
        ts += [t for t in (c.validity, c.data, c.offsets) if t is not None]
    return ts


def setup_workload(name, rows, args, rank=0):
    """Inputs generated in HBM, the operator made, outputs allocated and touched by one evaluation.
    Returns a namespace: step() = one pass of the hot path over the batch (what is timed)."""
    import types
    import torch
    import gandiva_amd as gandiva
    from gandiva_amd import workloads as W
    wl = types.SimpleNamespace(name=name, rows=rows, placeable=name in ("c1", "c2", "c4"), outs=None, out=None,
                               data_stream="BASELINE.md §4's distributions from torch's device generator (Philox), "
                                           "not its PCG64 stream")
    if name == "c2":
        gen = W.c2_device_batch_pcg64 if args.data == "pcg64" else W.c2_device_batch
        # every rank: its own shard (rank 0 = BASELINE.md §4's seeds, rank r = seeds + 1000 r)
        wl.dbatch = gen(rows, seed_offset=1000 * rank)
        if args.data == "pcg64":
            wl.data_stream = ("BASELINE.md §4: numpy PCG64, value seeds 42-45, mask seeds 142-145 (rank r: + 1000 r); "
                              "generated on the host, resident in HBM before the timed region")
        wl.obj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
        wl.bytes_per_row, wl.read_per_row = W.C2_BYTES_PER_ROW, 4 * 8 + 4 / 8
        wl.kernel_desc = "fused 10-expression projection kernel (1 launch per step)"
    elif name == "c1":
        g = torch.Generator(device="cuda")
        cols = []
        for k in range(3):
            g.manual_seed(1 + k)
            data = torch.empty(rows, dtype=torch.int32, device="cuda")
            data.random_(-(1 << 15), 1 << 15, generator=g)
            cols.append(gandiva.DeviceColumn(W.c1_schema().field(k).type, rows, None, data.view(torch.uint8)))
        wl.dbatch = gandiva.DeviceBatch(W.c1_schema(), cols, rows)
        wl.obj = gandiva.make_projector(W.c1_schema(), W.c1_expressions(), None)
        wl.bytes_per_row, wl.read_per_row = 16 + 1 / 8, 12
        wl.kernel_desc = "fused (a+b)*c int32 projection kernel"
    elif name == "c4":
        wl.dbatch = W.c4_device_batch(rows)
        wl.obj = gandiva.make_projector(W.c4_schema(), W.c4_expressions(), None)
        wl.bytes_per_row = 3 * 16 + 4 + 2 * 16 + 4 + 3 / 8  # inputs carry no validity buffers here
        wl.read_per_row = 3 * 16 + 4
        wl.kernel_desc = "fused decimal128 x2 + datediff projection kernel (1 launch per step)"
    elif name == "c5":
        wl.dbatch = W.c5_device_batch_philox(rows) if args.data_c5 == "philox" else W.c5_device_batch(rows)
        if args.data_c5 != "philox":
            wl.data_stream = "BASELINE.md §4: numpy PCG64 seed 21, generated on the host"
        wl.obj = gandiva.make_projector(W.c5_schema(), W.c5_expressions(), None)
        wl.kernel_desc = ("wave-shaped var-len plan: offsets-only pre-pass + offsets scan + main kernel of "
                          "independent wave tiles (byte sweep per 64-row sub-tile: match bits + LDS mirror of the span, "
                          "flat output from the sweep's registers, substr staged LDS -> LDS)")
    elif name == "c3":
        wl.dbatch = W.c3_device_batch(rows)
        wl.obj = gandiva.make_filter(W.c3_schema(), W.c3_condition())
        wl.out = torch.empty(rows, dtype=torch.int32, device="cuda")
        wl.kernel_desc = "predicate+ballot kernel, offsets scan (3 launches), index emit"
    elif name == "k2f":
        wl.dbatch = W.c3_device_batch(rows)
        wl.obj = gandiva.make_filter_project(W.c3_schema(), W.c3_condition(), W.c3_sum_expression(), "int32")
        if not wl.obj.fused:
            raise RuntimeError("the fused filter-project plan was not taken")
        wl.out = torch.empty(rows, dtype=torch.int32, device="cuda")
        wl.kernel_desc = ("ONE kernel: predicate, decoupled look-back over workgroup tiles, selected rows' a+b and "
                          "row indices staged in a wave-private LDS window and stored compacted (+ a one-thread count kernel)")
    else:
        raise SystemExit(f"unknown workload {name}")

    if name in ("c1", "c2", "c4", "c5"):
        wl.outs = wl.obj.evaluate_device(wl.dbatch)       # allocates + first touch
        wl.result = lambda: wl.outs

        def step():
            wl.obj.evaluate_device(wl.dbatch, outputs=wl.outs, sync=False)
        if name == "c5" and not args.c5_sync:
            # A var-len plan's synchronous entry reads the byte totals back before it returns: the GPU idles while the host
            # turns around (0.03-0.07 ms of a 1.05 ms step).  gdv_projector_evaluate_async enqueues pre-pass, scan and main
            # kernel and returns; status word and byte totals land in a device block, checked after the loop.
            wl.kernel_desc_suffix = "; step = gdv_projector_evaluate_async (no host wait: status word + byte totals stay on the device, checked after the timed loop)"

            def step():  # noqa: F811
                wl.outs, wl.last_result = wl.obj.evaluate_device_async(wl.dbatch, outputs=wl.outs)

            def result():
                torch.cuda.synchronize()
                st = int(wl.last_result[0])
                if st != 0:
                    raise RuntimeError(f"the asynchronous evaluations left status {st:#x}: outputs not complete")
                return wl.outs
            wl.result = result
        if name == "c5":
            outs = wl.outs
            in_bytes = 4 * (rows + 1) + int(sum(o.data_used for o in outs[2:]))  # offsets + data (upper preserves bytes)
            out_bytes = rows / 8 + sum(4 * (rows + 1) + o.data_used for o in outs[1:]) + 3 * rows / 8
            wl.bytes_per_row, wl.read_per_row = (in_bytes + out_bytes) / rows, in_bytes / rows
    elif name == "c3":
        sel = wl.obj.evaluate_device(wl.dbatch, "int32", out=wl.out)
        wl.bytes_per_row, wl.read_per_row = 16 + 4 * sel.num_slots / rows, 16

        def step():
            # (asynchronous, like the other workloads' steps: the slot count stays in device memory, where a selection-mode
            # Projector reads it — no host wait between steps, so a host thread that loses its core does not idle the GPU)
            wl.obj.evaluate_device(wl.dbatch, "int32", out=wl.out, sync=False)

        def result():
            wl.out.fill_(-1)       # the indices the check reads are written by THIS call, after the loop's
            return wl.obj.evaluate_device(wl.dbatch, "int32", out=wl.out)
        wl.result = result
    else:  # k2f
        wl.outs, sel = wl.obj.evaluate_device(wl.dbatch, indices=wl.out)
        share = sel.num_slots / rows
        wl.bytes_per_row, wl.read_per_row = 16 + (4 + 8 + 1 / 8) * share, 16

        def step():
            wl.obj.evaluate_device(wl.dbatch, outputs=wl.outs, indices=wl.out, sync=False)

        def result():
            wl.out.fill_(-1)
            wl.outs[0].data.fill_(0)
            return wl.obj.evaluate_device(wl.dbatch, outputs=wl.outs, indices=wl.out)
        wl.result = result
    wl.step = step
    # Tier 0 (round 6): a plan whose specialised code object is not in the cache yet starts on the interpreter kernel while
    # hipRTC runs in the background.  The bench measures the specialised kernel: wait for it (normally 0 s — build()
    # precompiles every BASELINE plan into gandiva_amd/_kcache), and say how long that took.
    from gandiva_amd import _capi
    lib, t0 = _capi.lib(), time.perf_counter()
    while time.perf_counter() - t0 < 60:
        before = lib.gdv_tier0_launches()
        step()
        torch.cuda.synchronize()
        if lib.gdv_tier0_launches() == before:
            break
        time.sleep(0.05)
    wl.tier0_wait_s = round(time.perf_counter() - t0, 2)
    wl.footprint = sum(t.numel() * t.element_size() for t in tensors_of(wl.dbatch, wl.outs)) + (
        wl.out.numel() * 4 if wl.out is not None else 0)
    return wl


def kernel_names_of(wl):
    import re
    seen = []
    for m in re.findall(r"gdv_k_[0-9a-f]{16}", wl.obj.llvm_ir or ""):
        if m not in seen:
            seen.append(m)
    return seen


def search_placements(wl, wanted):
    """Placement (round 5).  The same kernel on the same data runs 4.9 .. 7.0 ms per C2 step depending on WHERE the
    driver placed the buffers — a property of the allocation that stays with it (profiles/r05_box_states.txt).  The
    columns are copied into, and the outputs allocated as, fresh allocations up to `wanted` times; each placement is
    timed for 3 steps after 2 untimed, the fastest is kept, and EVERY trial's time goes into the line (the first entry
    = the first allocation, what a caller who allocates once gets).
    Round 6: memory is bounded — the best placement, the candidate and ONE loser are alive at a time (the loser stays
    until the next candidate exists, so that the driver cannot hand the same pages back), and the number of trials is
    cut to what torch.cuda.mem_get_info() says fits (C4: 66 GB per placement)."""
    import torch
    import gandiva_amd as gandiva
    if wanted <= 1 or not wl.placeable:
        return None

    def time_placement():
        for _ in range(2):
            wl.step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            wl.step()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 3
    for _ in range(40):   # (the first ~0.2 s after the generation are not representative of any placement)
        wl.step()
    trials = [round(time_placement(), 4)]
    best_ms, best, loser = trials[0], (wl.dbatch, wl.outs), None
    while len(trials) < wanted:
        free_b, _total = torch.cuda.mem_get_info()
        if free_b < wl.footprint * 1.05 + (2 << 30):
            if loser is None:
                break                     # not even one more placement fits next to the best one
            loser = None                  # give the loser's memory back first (its pages may come back: still a trial)
            torch.cuda.empty_cache()
            if torch.cuda.mem_get_info()[0] < wl.footprint * 1.05 + (2 << 30):
                break
        cols = [gandiva.DeviceColumn(c.type, c.length, None if c.validity is None else c.validity.clone(), c.data.clone())
                for c in best[0].columns]
        wl.dbatch = gandiva.DeviceBatch(best[0].schema, cols, wl.rows)
        wl.outs = wl.obj.evaluate_device(wl.dbatch)      # fresh output allocations
        del cols
        loser = None
        torch.cuda.empty_cache()
        t_ms = time_placement()
        trials.append(round(t_ms, 4))
        if t_ms < best_ms:
            loser, best_ms, best = best, t_ms, (wl.dbatch, wl.outs)
        else:
            loser = (wl.dbatch, wl.outs)
    wl.dbatch, wl.outs = best
    del best, loser
    torch.cuda.empty_cache()
    return trials


def time_steps(wl, warm, steps):
    import torch
    for _ in range(warm):
        wl.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        wl.step()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def pool_outputs(wl, candidates):
    """Round 6 (verdict item 2): the placement search belongs to the product.  The output columns of a projection workload
    are taken from the library's device pool: gdv_device_pool_reserve_set allocates four times the set as single buffers,
    forms candidates from every fourth one (members spread over a wide span of allocations: what is slow is a set of
    neighbours), times a non-temporal write sweep over each candidate (profiles/r06_placement_*.txt), keeps the fastest
    and returns everything else to the driver.  What a caller gets WITHOUT
    the pool — the first plain allocation — is timed first and reported next to it.  Returns the `placement` object."""
    import torch
    import gandiva_amd as gandiva
    if candidates <= 0 or not wl.placeable:
        return None
    first_ms = time_steps(wl, 40, 3)      # (the first ~0.2 s after the generation are not representative of any placement)
    wl.outs = None
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    wl.pool = gandiva.DevicePool()
    wl.outs = wl.pool.reserve_outputs(wl.obj, wl.rows, candidates)
    setup_ms = (time.perf_counter() - t0) * 1e3
    wl.step()                             # first touch of the kept set by the product kernel
    torch.cuda.synchronize()
    return {"by": "the library's device pool: gdv_device_pool_reserve_set (4x the output set allocated as single buffers, candidates = "
                  "every 4th one, probed with a non-temporal write sweep, fastest kept, everything else returned to the driver)",
            "sets": wl.pool.last_probe, "reserve_ms": round(setup_ms, 1),
            "first_plain_allocation_ms": round(first_ms, 4)}


def timed_loop(wl, steps, warmup, barrier=None, prewarm_s=0.4, sampler=None):
    """Steady state before anything is counted: the GPU's clocks and power state are still moving for the first few
    hundred milliseconds of work (round 4, one box, same kernel, same buffers: 5.58 ms per step in the first second after
    the inputs were generated, 4.90 ms a second later).  ~prewarm_s of untimed steps come first (config.prewarm_steps),
    THEN the W warm-up steps, then EXACTLY K timed ones between two barriers; HIP events around every step."""
    import torch
    torch.cuda.synchronize()
    t_probe = time.perf_counter()
    wl.step()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t_probe, 1e-5)
    prewarm = 0 if os.environ.get("GDV_BENCH_NO_PREWARM") else max(10, min(2000, int(prewarm_s / one)))
    for _ in range(prewarm):
        wl.step()
    torch.cuda.synchronize()
    for _ in range(warmup):
        wl.step()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    if barrier:
        barrier()
    else:
        torch.cuda.synchronize()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for i in range(steps):
        starts[i].record()
        wl.step()
        ends[i].record()
    if barrier:
        barrier()
    else:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if sampler:
        sampler.stop()
    return elapsed, [s.elapsed_time(e) for s, e in zip(starts, ends)], prewarm


def verify(wl, no_verify):
    if no_verify:
        return {"ok": None, "what": "skipped (--no-verify)"}
    try:
        return verify_outputs(wl.name, wl.rows, wl.dbatch, wl.result())
    except Exception as e:  # a check that cannot run is a failed check
        return {"ok": False, "what": f"verification raised {type(e).__name__}: {e}"}


def roofline_of(wl, dev_ms, trials, quote_traffic, placement=None):
    mean_ms = sum(dev_ms) / len(dev_ms)
    achieved = wl.bytes_per_row * wl.rows / (mean_ms * 1e-3) / 1e9
    names = kernel_names_of(wl)
    traffic, source = (load_traffic(wl.name, names[0] if names else None) if quote_traffic
                       else (None, "PMC passes are taken at the BASELINE size only"))
    r = {
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_source": source,
        "kernel_name": names[0] if names else None,
        "kernel_names": names + AOT_KERNELS.get(wl.name, []),
        "kernel": wl.kernel_desc + getattr(wl, "kernel_desc_suffix", ""),
        "algorithmic_bytes_per_row": round(wl.bytes_per_row, 3),
        # the read and write shares of `achieved`, separately
        "read_bytes_per_row": round(wl.read_per_row, 3),
        "write_bytes_per_row": round(wl.bytes_per_row - wl.read_per_row, 3),
        "achieved_read": round(achieved * wl.read_per_row / wl.bytes_per_row, 1),
        "achieved_write": round(achieved * (1 - wl.read_per_row / wl.bytes_per_row), 1),
        "kernel_ms": round(mean_ms, 4),
        "kernel_ms_min": round(min(dev_ms), 4),
        "kernel_ms_max": round(max(dev_ms), 4),
        # (`frac` is on the MEAN of the timed steps, as the contract asks; the median next to it shows when one step of a
        # short loop was stretched — a synchronous Evaluate launches from the host several times per step, and a host
        # thread that loses its core for a few milliseconds leaves the GPU idle between two of those launches)
        "kernel_ms_median": round(sorted(dev_ms)[len(dev_ms) // 2], 4),
        **({"kernel_ms_steps": [round(x, 3) for x in dev_ms]} if os.environ.get("GDV_BENCH_STEP_TIMES") else {}),
        # ms per step of every placement tried before the timed loop (first = the first allocation); the
        # timed loop ran on the fastest.  null: one allocation, no trials
        "placement_trials_ms": trials,
    }
    if placement:
        # where the timed loop's output columns came from, and what the first plain allocation of this process ran at
        r["placement"] = placement
        r["frac_first_plain_allocation"] = round(wl.bytes_per_row * wl.rows / 1e9 / (placement["first_plain_allocation_ms"] * 1e-3) / HBM_PEAK_GBS, 4)
    if trials:
        # what a caller who allocates ONCE gets (first entry), and the middle of what this box offered
        alg = wl.bytes_per_row * wl.rows / 1e9
        med = sorted(trials)[len(trials) // 2] if len(trials) % 2 else sum(sorted(trials)[len(trials) // 2 - 1: len(trials) // 2 + 1]) / 2
        r["frac_first_allocation"] = round(alg / (trials[0] * 1e-3) / HBM_PEAK_GBS, 4)
        r["frac_median_placement"] = round(alg / (med * 1e-3) / HBM_PEAK_GBS, 4)
        r["frac_worst_placement"] = round(alg / (max(trials) * 1e-3) / HBM_PEAK_GBS, 4)
    return r, achieved, mean_ms


def short_cpu_baseline(name, budget_s):
    """The oracle ("port") on this host's cores on a bounded sample of the same workload (a few seconds)."""
    from gandiva_amd import workloads as W
    from oracle import oracle
    host = host_description()
    cores = host["usable_cores"]
    if name == "c1":
        rows = 1 << 24
        batch, exprs = W.c1_batch(rows), W.c1_expressions()
        outs = oracle.alloc_outputs(exprs, rows)
        fn, thr = (lambda: oracle.project(exprs, batch, threads=cores, out=outs)), cores
    elif name == "c3":
        rows = 1 << 25
        batch, cond = W.c3_batch(rows), W.c3_condition()
        fn, thr = (lambda: oracle.filter_indices(cond, batch, "int32", threads=cores)), cores
    elif name == "k2f":
        rows = 1 << 25
        batch, cond, exprs = W.c3_batch(rows), W.c3_condition(), W.c3_sum_expression()

        def fn():
            idx = oracle.filter_indices(cond, batch, "int32", threads=cores)
            oracle.project(exprs, oracle.take_rows(batch, idx.to_numpy()), threads=cores)
        thr = cores
    elif name == "c4":
        rows = 1 << 22
        batch, exprs = W.c4_batch(rows), W.c4_expressions()
        fn, thr = (lambda: oracle.project(exprs, batch, threads=cores)), cores
    elif name == "c5":
        rows = 1 << 20
        batch, exprs = W.c5_batch(rows), W.c5_expressions()
        fn, thr = (lambda: oracle.project(exprs, batch)), 1
    else:
        raise KeyError(name)
    fn()
    rate, reps, el = timed_passes(fn, rows, budget_s, 200)
    return {"value": round(rate / 1e6, 2), "unit": "million rows/s", "cores": thr, "kind": "port",
            "sample": f"{reps} passes x {rows} rows, {thr} thr, {el:.1f}s", "cpu_model": host["cpu_model"],
            "loadavg": host["loadavg"]}


def sub_line(name, args):
    """One BASELINE config as a sub-object of the bench line: its own inputs, timed loop (HIP events per step),
    verification against torch and a short CPU baseline.  Everything it allocated is released on return."""
    import torch
    t_wall = time.perf_counter()
    rows = DEFAULT_ROWS[name]
    wl = setup_workload(name, rows, args)
    placement = pool_outputs(wl, min(args.pool_candidates, args.sub_placements))
    trials = search_placements(wl, min(args.placements, args.sub_placements))
    elapsed, dev_ms, prewarm = timed_loop(wl, args.sub_steps, args.sub_warmup, prewarm_s=0.25)
    verification = verify(wl, args.no_verify)
    roof, achieved, mean_ms = roofline_of(wl, dev_ms, trials, True, placement)
    metric, dtype, text = WORKLOAD_TEXT[name]
    line = {
        "metric": metric,
        "value": round(rows * args.sub_steps / elapsed / 1e6, 1),
        "unit": "million rows/s",
        "steps": args.sub_steps,
        "warmup": args.sub_warmup,
        "ms_per_step": round(elapsed / args.sub_steps * 1e3, 4),
        "kernel_ms": round(mean_ms, 4),
        "dtype": dtype,
        "verified": None if args.no_verify else bool(verification["ok"]),
        "verification": verification,
        "config": {"workload": text, "rows": rows, "prewarm_steps": prewarm, "data_stream": wl.data_stream,
                   "waited_for_the_specialised_kernel_s": wl.tier0_wait_s,
                   "residency": "inputs and outputs in HBM (zero-copy C-ABI path)"},
        "roofline": roof,
    }
    del wl
    torch.cuda.empty_cache()
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = short_cpu_baseline(name, args.sub_cpu_seconds)
        except Exception as e:  # the baseline must never take the bench line down
            line["cpu_baseline"] = {"value": None, "unit": "million rows/s", "cores": 0, "kind": "port",
                                    "sample": f"failed: {e}"}
    line["wall_s"] = round(time.perf_counter() - t_wall, 1)
    return line


def main():
    args = parse()
    if args.inproc:
        return main_inproc(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: gandiva_amd has no CPU evaluation path")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the two must agree")
    # GDV_BENCH_BACKEND=gloo lets the N>1 control flow (rendezvous, barrier, max-over-ranks
    # aggregation) be exercised on a single-GPU box with several ranks sharing cuda:0; the
    # driver's multi-GPU runs use the default, nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("GDV_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: this node exposes {torch.cuda.device_count()} GPU(s)")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    reduce_device = "cuda" if backend == "nccl" else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # rows of this rank.  weak: --rows (default: the BASELINE size) PER GPU.  strong: ONE logical batch of the
    # BASELINE size (C2 2^28, C3 10^9, C4 6*10^9 = 750 M per GPU on 8) row-sharded on 1024-row boundaries: this
    # rank generates and evaluates only its own range (gdv_shard_bounds; no collective on the data path).
    strong = args.scaling == "strong"
    logical_rows = None
    rows = args.rows or DEFAULT_ROWS[args.workload]
    if strong:
        from gandiva_amd import shard
        logical_rows = args.rows or STRONG_ROWS[args.workload]
        lo, hi = shard.shard_bounds(logical_rows, world, rank)
        rows = max(hi - lo, 1)
        if args.workload == "c4" and rows * 89 > 0.9 * torch.cuda.get_device_properties(device_index).total_memory:
            raise SystemExit(f"bench.py --workload c4 --scaling strong: {logical_rows} rows over {world} GPU(s) = {rows} rows "
                             f"({rows * 89 / 1e9:.0f} GB) per GPU do not fit its HBM; BASELINE's configuration is 8 GPUs "
                             f"(750 M rows each) — pass --rows for a smaller logical batch")
    wl = setup_workload(args.workload, rows, args, rank)
    placement = pool_outputs(wl, args.pool_candidates)
    placement_trials = search_placements(wl, args.placements)

    try:
        bus = torch.cuda.get_device_properties(device_index).pci_bus_id
        bus_id = f"{int(bus):02x}:" if isinstance(bus, int) else str(bus)
    except Exception:
        bus_id = None
    sampler = BoxSampler(bus_id if rank == 0 else None) if rank == 0 and os.environ.get("GDV_BENCH_NO_TELEMETRY") is None else None
    if sampler is not None and sampler.dev is None:
        sampler = BoxSampler(None)   # bus id did not match a sysfs path: take the first amdgpu card
    elapsed, dev_ms, prewarm = timed_loop(wl, args.steps, args.warmup, barrier=barrier, sampler=sampler)

    # what did the timed loop leave behind?  (outside the timed region; every rank checks its own shard)
    verification = verify(wl, args.no_verify)
    bad = 0.0 if verification["ok"] in (True, None) else 1.0

    roof, achieved, mean_dev_ms = roofline_of(wl, dev_ms, placement_trials, not args.rows and not strong, placement)
    t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
    k = torch.tensor([mean_dev_ms], dtype=torch.float64, device=reduce_device)
    f = torch.tensor([bad], dtype=torch.float64, device=reduce_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
    any_rank_failed = float(f.item()) > 0
    elapsed = float(t.item())
    if float(k.item()) != mean_dev_ms:        # the slowest rank's kernel time prices the roofline
        mean_dev_ms = float(k.item())
        achieved = wl.bytes_per_row * rows / (mean_dev_ms * 1e-3) / 1e9
        roof.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "kernel_ms": round(mean_dev_ms, 4)})

    if rank == 0:
        total_rows = logical_rows if strong else rows * world
        metric, dtype, text = WORKLOAD_TEXT[args.workload]
        value = total_rows * args.steps / elapsed / 1e6
        line = {
            "metric": metric,
            "value": round(value, 1),
            "unit": "million rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "verified": (None if args.no_verify else not any_rank_failed),
            "verification": verification if not any_rank_failed or not verification["ok"] else
                            {"ok": False, "what": "another rank's check failed"},
            "config": {
                "workload": text,
                "rows_per_gpu": rows,
                "total_rows": total_rows,
                "prewarm_steps": prewarm,
                "waited_for_the_specialised_kernel_s": wl.tier0_wait_s,
                "data_stream": wl.data_stream,
                "sharding": f"row-range x{world}, no collective" + (
                    f"; strong: ONE logical batch of {logical_rows} rows, shard r = rows [r n/N, (r+1) n/N) on 1024-row bounds" if strong else
                    "; weak: every rank evaluates its own batch of rows_per_gpu rows"),
                "residency": "inputs and outputs in HBM (zero-copy C-ABI path)",
            },
            "roofline": roof,
        }
        # the box this number was taken on: telemetry during the timed loop + what plain streaming
        # kernels reach here, now, in this process
        box = sampler.summary() if sampler else {}
        ceil = None
        try:
            torch.cuda.synchronize()
            ceil = hbm_ceilings()
        except Exception as e:  # never take the bench line down
            box["ceiling_error"] = str(e)
        shaped = None
        try:
            if args.workload == "c2":
                # the headline: the ceiling is taken on the very buffers the timed loop read and wrote
                shaped = stream_ceiling_on([c.data for c in wl.dbatch.columns], [o.data for o in wl.outs], rows)
            else:
                nr, nw = CEILING_SHAPE[args.workload]
                per_stream = max(1 << 28, min(1 << 32, int(wl.bytes_per_row * rows / (nr + nw)) & ~8191))
                shaped = stream_ceiling(nr, nw, per_stream)
        except Exception as e:
            box["ceiling_error"] = str(e)
        if ceil:
            box["ceiling"] = ceil
            rshare = wl.read_per_row / wl.bytes_per_row
            mixed = 1.0 / (rshare / ceil["read"] + (1.0 - rshare) / ceil["write"])
            box["ceiling_for_this_read_write_mix"] = round(mixed, 1)
            # the ceiling = the best ANY of the streaming instruments reaches for this traffic: the
            # shape-matched kernel (same number of input / output streams as the workload, grid and
            # non-temporal flag swept) or the one-stream read / write rates combined for the mix
            best = max(mixed, shaped["GB/s"] if shaped else 0.0)
            if shaped:
                box["ceiling_same_shape"] = shaped
            if achieved > best:
                # the product kernel outran every streaming instrument of this run (a fraction of a percent, run-to-run
                # noise between two kernels of the same shape): the ceiling is then what the product reached
                box["ceiling_raised_to_the_product_kernel"] = True
                best = achieved
            line["roofline"]["measured_ceiling"] = round(best, 1)
            line["roofline"]["frac_of_measured_ceiling"] = round(achieved / best, 4)
        line["roofline"]["box"] = box
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.workload == "c2":
                    line["cpu_baseline"] = cpu_baseline_c2(args.cpu_rows)
                else:
                    line["cpu_baseline"] = short_cpu_baseline(args.workload, 8.0)
            except Exception as e:  # the baseline must never take the bench line down
                line["cpu_baseline"] = {"value": None, "unit": "million rows/s", "cores": 0,
                                        "kind": "port", "sample": f"failed: {e}"}
        # Round 6: the other BASELINE configs (and the fused filter -> project) in the SAME line, so that the driver's
        # one run times them all: "workloads": {c1, c3, k2f, c5, c4} — each with ms_per_step, kernel_ms (HIP events),
        # kernel names, roofline.frac on §8(d)'s algorithmic bytes, PMC traffic where a pass on the running kernel is
        # committed, verified, and a short cpu_baseline.  Only for the default invocation (one GPU, C2, BASELINE size).
        if world == 1 and args.workload == "c2" and not args.rows and not strong and not args.no_extras:
            del wl
            torch.cuda.empty_cache()
            subs = {}
            for name in args.extras.split(","):
                try:
                    subs[name] = sub_line(name, args)
                except Exception as e:  # one workload's failure is reported, not fatal to the headline
                    subs[name] = {"verified": False, "error": f"{type(e).__name__}: {e}"}
                    torch.cuda.empty_cache()
            line["workloads"] = subs
            if not args.no_verify and not all(v.get("verified") for v in subs.values()):
                any_rank_failed = True
                verification = {"ok": False, "what": "a sub-workload failed: " + ", ".join(
                    k for k, v in subs.items() if not v.get("verified"))}
                line["verified"] = False
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if any_rank_failed:
        raise SystemExit("bench.py: the outputs of the timed loop failed verification: " + str(verification))


if __name__ == "__main__":
    main()
