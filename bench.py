#!/usr/bin/env python3
"""Headline benchmark: demodulated stream samples per second on BASELINE config 2.

One "step" = one pass of the receive hot path over one batch of frame buffers that
is already resident in HBM:
    gf3_sync_frames  (chirp matched filter + peak rule, window 0..320 lags)
 -> gf3_demod_frames (CP strip + FFT + pilot LS + phase slope + equalise + demap + bit-pack)
 -> (N>1 only) one RCCL all-gather of the bit-packed payload.
Workload: N=4096, CP=512, P=2, D=8, QPSK on bins 1..2046, F frames per GPU (weak
scaling), each frame a row of `stride` fp32 samples = [jitter gap 0..299 | chirp |
2 pilots | 8 data | 2 pilots | pad], all distinct, synthesised on device by gf3_tx_frames.

    python bench.py --gpus 1 --steps 100 --warmup 5      (the defaults: a timed region of ~0.8 s)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...      (not under torchrun: starts its own N rank processes, see launch_ranks)

N = 1 is BASELINE config 2 (65 536 frames); N > 1 is BASELINE config 4 (131 072 frames per GPU = 1 048 576 at
N = 8, weak scaling), timed with the per-chunk overlapped all-gather (`value`) and again with the literal single
all-gather of the north_star (`single_gather`).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for
the dominant kernel (demod_kernel) and `cpu_baseline` (the NumPy oracle, 1 thread, on
a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WIN_LO = -8                    # search window = chirp-start lags [-8, window-8): jitter gaps 0..299 are interior points


class PowerSampler:
    """Package power of this process's GPU while the timed loop runs, read from the amdgpu hwmon node in sysfs
    (plain file reads from a thread: nothing is executed).  median_w is None when the node cannot be found."""

    def __init__(self, dev_index):
        import glob, threading
        self.path, self.samples, self._stop = None, [], threading.Event()
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for card in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(card).endswith(bdf):
                    hits = glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")) + \
                           glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average"))
                    if hits:
                        self.path = hits[0]
        except Exception:
            self.path = None
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(int(open(self.path).read()) * 1e-6)
            except Exception:
                pass
            self._stop.wait(0.02)

    def __enter__(self):
        if self.path:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.path:
            self._thread.join()

    @property
    def median_w(self):
        return float(np.median(self.samples)) if self.samples else None


def build_workload(args, rank, N=4096, frames=None, stride=None):
    """F distinct frame buffers, synthesised on the device by the engine's own transmit kernel
    (gf3_tx_frames): random payload, jitter gap 0..299 before each chirp, QPSK filler on the one
    non-data carrier.  Returns the rows and what the receiver must recover.  (N, frames, stride: the same
    geometry -- CP = N/8, P = 2, D = 8, QPSK on bins 1..K-1 -- at another symbol size, for the by-N legs.)"""
    from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table
    CP, P, D = N // 8, 2, 8
    K = N // 2 - 1
    frames = frames or args.frames
    stride = stride or args.stride
    pts, bt = qpsk_table()
    known = np.unpackbits(np.load(os.path.join(ROOT, "gf3_audio_modem_amd", "data", "known_bits.npz"))["packed"])
    known = np.tile(known, -(-K * 2 // len(known)))
    cfg = RxConfig(N=N, CP=CP, P=P, D=D, data_bins=np.arange(1, K), const_points=pts, const_bits=bt,
                   known_bits=known, in_dtype=torch.float32, max_window=args.window)
    eng = Engine(cfg)
    gen = torch.Generator(device="cuda").manual_seed(20261003 + rank)
    payload = torch.randint(0, 256, (frames, eng.bytes_per_frame), dtype=torch.uint8, device="cuda", generator=gen)
    gaps = torch.randint(0, 300, (frames,), dtype=torch.int64, device="cuda", generator=gen)
    filler = np.zeros(K, dtype=complex)
    filler[K - 1] = pts[(20261003 + rank) % 4]                  # random_qpsk for the unused carrier (OFDM.py:201-215)
    big = eng.tx_frames(payload, filler, stride=stride, gaps=gaps, out_dtype=torch.float32)
    return eng, cfg, big, payload, gaps


def by_N_rooflines(dev, args, reps=20):
    """The two fused kernels of the headline step at the other symbol sizes SURVEY section 2 generalises to
    (OFDM.py:27 hard-codes 4096): N in {1024, 2048, 8192} with the headline's geometry scaled (CP = N/8, P = 2, D = 8,
    QPSK on bins 1..K-1, chirp 5 (N + CP), f32 samples, jitter gaps 0..299, 320-lag window) and the SAME number of
    stream samples per launch -- frames = 65 536 x 4096 / N.  Each kernel alone: median of 20 launches after 3 (HIP
    events on the launch stream); every sync offset and every payload byte of the timed output checked.
    Algorithmic bytes as SURVEY 8(d): demod B_in M N + D C mu / 8 per packet; sync B_in (Lc + W - 1) + 8."""
    out = {"roofline_demod_by_N": {}, "roofline_sync_by_N": {}}
    for N in (1024, 2048, 8192):
        frames = max(64, args.frames * 4096 // N)
        S = N + N // 8
        stride = (17 * S + 320 + 63) // 64 * 64                  # chirp + 12 symbols + the largest gap, rounded up
        eng, cfg, big, payload, gaps = build_workload(args, 0, N=N, frames=frames, stride=stride)
        exp = torch.arange(frames, device=dev, dtype=torch.int64) * stride + gaps + cfg.chirp_length
        starts = torch.empty((frames,), dtype=torch.int64, device=dev)
        bits = torch.empty((frames, eng.bytes_per_frame), dtype=torch.uint8, device=dev)
        ms_sync = _event_ms(lambda: eng.sync_frames(big, frames, stride, WIN_LO, WIN_LO + args.window, out_starts=starts), reps)
        ms_dem = _event_ms(lambda: eng.demod_frames(big, starts, out_bits=bits, split=False), reps)
        ok_sync, ok_bits = bool(torch.equal(starts, exp)), bool(torch.equal(bits, payload))
        by_d = frames * (4 * cfg.M * N + eng.bytes_per_frame)
        by_s = frames * (4 * (cfg.chirp_length + args.window - 1) + 8)
        common = {"frames_per_launch": frames, "samples_per_launch": frames * stride, "timing": f"median of {reps} launches after 3 warm-ups"}
        out["roofline_demod_by_N"][f"N{N}"] = dict(common, kernel=f"demod_kernel<{N // 2},f32,MODE_QPSK> ({N // 16} threads per packet)", bound="hbm",
                                                  achieved=by_d / ms_dem / 1e6, peak=HBM_PEAK_GBS, unit="GB/s", frac=by_d / ms_dem / 1e6 / HBM_PEAK_GBS,
                                                  algorithmic_bytes_per_launch=by_d, avg_launch_ms=ms_dem, payload_recovered=ok_bits)
        out["roofline_sync_by_N"][f"N{N}"] = dict(common, kernel="corr_kernel (plan chosen by the context for this N)", bound="hbm",
                                                 achieved=by_s / ms_sync / 1e6, peak=HBM_PEAK_GBS, unit="GB/s", frac=by_s / ms_sync / 1e6 / HBM_PEAK_GBS,
                                                 algorithmic_bytes_per_launch=by_s, avg_launch_ms=ms_sync, sync_offsets_exact=ok_sync)
        del big, payload, gaps, starts, bits, exp
        eng.close()
        torch.cuda.empty_cache()
    return out


FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: fp64 vector (= matrix) peak at 2.4 GHz


def power_roofline(dev_index, fn, t_launch_s, F, lib_sha16, seconds=2.0):
    """What actually bounds the dominant kernel: the package power cap.  `fn` (one launch of demod_kernel over
    the headline batch) is repeated ALONE for `seconds` while the amdgpu hwmon node is sampled -> measured_w and
    measured_uJ_per_packet of THIS run; the cap is read from the same node.  The model beside it is the energy budget of
    tools/energy_model.py -- resident-grid power x time + sum over instruction classes of count x energy per operation +
    HBM bytes x energy per byte -- evaluated at this run's launch time from the committed microbenchmark prices
    (profiles/r03_energy_budget.json) and the committed instruction mix of the kernel (profiles/instruction_mix_current.json,
    used only if it was collected on the sources the loaded library was built from)."""
    import importlib.util
    ps = PowerSampler(dev_index)
    cap_w = None
    try:
        if ps.path:
            cap_w = int(open(os.path.join(os.path.dirname(ps.path), "power1_cap")).read()) * 1e-6
    except Exception:
        cap_w = None
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = 0
    with ps:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                fn()
            torch.cuda.synchronize(); n += 10
        dt = time.perf_counter() - t0
    w = float(np.median(ps.samples[len(ps.samples) // 2:])) if ps.samples else None
    out = {"what": "the dominant kernel is bound by the package power cap, not by bytes or issue slots (DESIGN 8.0): time = energy per launch / (cap - resident power)",
           "cap_w": cap_w, "measured_w": w, "kernel_alone_ms_per_launch": dt / n * 1e3, "launches_sampled": n,
           "measured_uJ_per_packet": (w * dt / n / F * 1e6) if w else None,
           "resident_w": None, "model_uJ_per_packet": None, "fp64_flop_per_launch": None, "fp64_tflops": None, "fp64_frac_of_78.6": None}
    try:
        spec = importlib.util.spec_from_file_location("gf3_energy_model", os.path.join(ROOT, "tools", "energy_model.py"))
        em = importlib.util.module_from_spec(spec); spec.loader.exec_module(em)
        eb = json.load(open(os.path.join(ROOT, "profiles", "r03_energy_budget.json")))
        mixf = json.load(open(os.path.join(ROOT, "profiles", "instruction_mix_current.json")))
        resident, e, st = em.slopes(eb)
        out["resident_w"] = resident
        out["prices_source"] = "profiles/r03_energy_budget.json (tools/ubench/energy_budget.py: slopes of package power over issue rate; streaming reads)"
        if mixf.get("source_sha16") != lib_sha16:
            out["model_note"] = f"profiles/instruction_mix_current.json was collected on sources {mixf.get('source_sha16')}, the loaded library is {lib_sha16}: no model"
            return out
        mix = next(v for k, v in mixf["kernels"].items() if k.startswith("demod_kernel<2048, 1, false, 2") and f"grid={F * 256}" in k)
        nbytes = F * 200700
        t_s = dt / n
        _, lo = em.budget(mix, t_s, nbytes, resident, e, st, False)
        _, hi = em.budget(mix, t_s, nbytes, resident, e, st, True)
        flop = 64.0 * (2 * mix["SQ_INSTS_VALU_FMA_F64"] + mix["SQ_INSTS_VALU_ADD_F64"] + mix["SQ_INSTS_VALU_MUL_F64"] + mix["SQ_INSTS_VALU_TRANS_F64"])
        out.update({"model_uJ_per_packet": [lo / F * 1e6, hi / F * 1e6],
                    "model_over_measured": [lo / (w * t_s), hi / (w * t_s)] if w else None,
                    "model_source": "tools/energy_model.py on profiles/instruction_mix_current.json (SQ_INSTS_* passes of this command, same source_sha16) "
                                    "x the committed prices; the bracket is structured ... random operand bits",
                    "fp64_flop_per_launch": flop, "fp64_tflops": flop / t_launch_s / 1e12,
                    "fp64_frac_of_78.6": flop / t_launch_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                    "model_ms_per_launch_at_cap": [(x - resident * t_s) / (cap_w - resident) * 1e3 for x in (lo, hi)] if cap_w else None,
                    "model_ms_note": "modelled dynamic energy of a launch / (cap - resident power): the launch time the cap allows"})
    except Exception as ex:                                      # (missing committed files: the measured half stands alone)
        out["model_note"] = f"model not evaluated: {type(ex).__name__}: {ex}"
    return out


_POOL_P = None


def _pool_init():
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass


def _pool_rows(job):
    """worker: oracle on a block of frame buffers -> packed bits (the parameters arrive once per job; cheap)"""
    from oracle import gf3_oracle as orc
    rows32, pk, window = job
    p = orc.RxParams(**pk)
    return np.packbits(orc.receive_rows(rows32.astype(np.float64), p, WIN_LO, WIN_LO + window)["bits"].astype(np.uint8).reshape(len(rows32), -1), axis=1)


def make_cpu_pool():
    """Fork the all-cores worker pool BEFORE anything touches the GPU (a forked child of a GPU-initialised
    process must not exist on this pool's boxes); the workers sleep until cpu_baseline_all_cores feeds them."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                  # the container's CPU share, when a cgroup quota sets one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("GF3_CPU_WORKERS", "16")))   # a GPU box grants 16 cores per GPU
    return mp.get_context("fork").Pool(cores, initializer=_pool_init), cores


def cpu_baseline_all_cores(pool, cores, cfg, rows_dev, payload_dev, window, max_frames=4096):
    """The same oracle on every host core the process may use (one worker per core, 16 frame buffers per job)."""
    n = int(min(rows_dev.shape[0], max_frames))
    rows = rows_dev[:n].cpu().numpy()                                   # float32, as stored
    pk = dict(N=cfg.N, CP=cfg.CP, P=cfg.P, D=cfg.D, lo=1, hi=cfg.K, const_points=cfg.const_points,
              const_bits=np.asarray(cfg.const_bits, dtype=np.int64), known_bits=cfg.known_bits)
    jobs = [(rows[s: s + 16], pk, window) for s in range(0, n, 16)]
    pool.map(_pool_rows, jobs[:cores])                                  # warm the workers (imports, page faults)
    t = time.perf_counter()
    out = pool.map(_pool_rows, jobs, chunksize=1)
    dt = time.perf_counter() - t
    ok = bool(np.array_equal(np.concatenate(out)[:, : payload_dev.shape[1]], payload_dev[:n].cpu().numpy()))
    return {"value": n * rows.shape[1] / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} config-2 frame buffers through oracle.receive_rows in a {cores}-process pool "
                      f"(job hand-over included), {dt:.1f} s, payload recovered: {ok}"}


def cpu_baseline(cfg, rows_dev, payload_dev, window, target_s):
    """Oracle ('port' of the reference algorithm, NumPy, 1 thread) on a bounded sample of
    the same frame buffers."""
    from oracle import gf3_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    p = orc.RxParams(N=cfg.N, CP=cfg.CP, P=cfg.P, D=cfg.D, lo=1, hi=cfg.K, const_points=cfg.const_points,
                     const_bits=np.asarray(cfg.const_bits, dtype=np.int64), known_bits=cfg.known_bits)
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        rows4 = rows_dev[:4].cpu().numpy().astype(np.float64)
        t = time.perf_counter(); orc.receive_rows(rows4, p, WIN_LO, WIN_LO + window); t4 = time.perf_counter() - t
        n = int(max(4, min(rows_dev.shape[0], target_s / (t4 / 4))))
        rows = rows_dev[:n].cpu().numpy().astype(np.float64)       # the first n of the GPU batch's frame buffers
        t = time.perf_counter()
        bits = []
        for s in range(0, n, 64):                                 # 64 rows at a time keeps the working set small
            bits.append(orc.receive_rows(rows[s: s + 64], p, WIN_LO, WIN_LO + window)["bits"])
        dt = time.perf_counter() - t
        bits = np.concatenate(bits)
    finally:
        if ctx is not None and hasattr(ctx, "unregister"):
            ctx.unregister()
    want = np.unpackbits(payload_dev[:n].cpu().numpy(), axis=1)[:, : cfg.bits_per_frame]
    ok = bool(np.array_equal(bits.reshape(n, -1), want))
    return {"value": n * rows.shape[1] / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{n} config-2 frame buffers ({n * rows.shape[1]} samples) through oracle.receive_rows "
                      f"(NumPy restatement of OFDM.py, 1 thread), {dt:.1f} s, payload recovered: {ok}"}


def _pool_stream(job):
    """worker: oracle.receive on one whole stream (stream-mode sync + demod) -> number of decoded bits"""
    from oracle import gf3_oracle as orc
    r, pk = job
    return int(len(orc.receive(r, orc.RxParams(**pk))["bits"]))


def _config1_stream():
    """BASELINE config 1 regenerated from the seed the g1b fixture records: 64 QPSK frames, N=1024, CP=128 (oracle.tx_stream
    rebuilds the reference's transmit() stream bit for bit, tests/test_oracle_golden.py)."""
    from oracle import gf3_oracle as orc
    g = np.load(os.path.join(ROOT, "tests", "golden", "g1b_config1_64f.npz"))
    pts, bt = orc.qpsk_table()
    pk = dict(N=int(g["N"]), CP=int(g["CP"]), P=int(g["P"]), D=int(g["D"]), lo=int(g["lo"]), hi=int(g["hi"]),
              const_points=pts, const_bits=bt, known_bits=g["known_bits"])
    p = orc.RxParams(**pk)
    F = int(g["F"])
    payload = np.random.RandomState(20261003).randint(0, 2, F * p.D * p.C * p.mu)
    r = orc.tx_stream(payload, g["fill"], p, gaps=g["gaps"], lead=int(g["lead"]), tail=int(g["tail"]))
    return r, pk, payload, "BASELINE config 1: 64 QPSK frames, N=1024 CP=128 P=2 D=8, one stream, noiseless"


def _config3_stream(F=24):
    """A 24-packet slice of BASELINE config 3: 16-QAM, N=4096 CP=512, through the measured 30-tap channel + noise."""
    from scipy.signal import lfilter
    from oracle import gf3_oracle as orc
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3_n4096_16qam_gr5.npz"))
    pk = dict(N=int(g["N"]), CP=int(g["CP"]), P=int(g["P"]), D=8, lo=int(g["lo"]), hi=int(g["hi"]),
              const_points=g["const_points"], const_bits=g["const_bits"].astype(np.int64), known_bits=g["known_bits"].astype(np.uint8))
    p = orc.RxParams(**pk)
    rs = np.random.RandomState(33)
    payload = rs.randint(0, 2, F * p.D * p.C * p.mu)
    r = orc.tx_stream(payload, g["fill"], p, gaps=rs.randint(0, 300, F), lead=50, tail=40)
    r = lfilter(g["channel"], 1.0, r) + 2e-4 * rs.randn(len(r))
    return r, pk, payload, f"{F}-packet slice of BASELINE config 3: 16-QAM, N=4096 CP=512 P=2 D=8, gr5channel.csv FIR + noise, one stream"


def cpu_baseline_streams(pool, cores, target_s=4.0):
    """the oracle on configs 1 and 3 next to config 2's -- whole-stream receive (matched filter over the
    stream, global-max peak rule, demod), 1 thread, and all cores as `cores` independent copies of the stream at once
    (the reference has no parallelism inside a stream)."""
    from oracle import gf3_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    out = {}
    for key, make in (("cpu_baseline_config1", _config1_stream), ("cpu_baseline_config3", _config3_stream)):
        r, pk, payload, what = make()
        p = orc.RxParams(**pk)
        ctx = threadpool_limits(limits=1) if threadpool_limits else None
        try:
            t = time.perf_counter(); res = orc.receive(r, p); t1 = time.perf_counter() - t       # (also the warm-up)
            reps = int(max(1, min(50, target_s / t1)))
            t = time.perf_counter()
            for _ in range(reps):
                res = orc.receive(r, p)
            dt = (time.perf_counter() - t) / reps
        finally:
            if ctx is not None and hasattr(ctx, "unregister"):
                ctx.unregister()
        ber = float(np.mean(res["bits"] != payload)) if len(res["bits"]) == len(payload) else None
        ent = {"value": len(r) / dt, "unit": "samples/s", "cores": 1, "kind": "port",
               "sample": f"{what}: {len(r)} samples through oracle.receive, mean of {reps} passes, {dt:.3f} s each, BER vs payload {ber}"}
        if pool is not None:
            jobs = [(r, pk)] * cores
            pool.map(_pool_stream, jobs[: cores])                                                 # warm the workers
            t = time.perf_counter(); pool.map(_pool_stream, jobs, chunksize=1); dta = time.perf_counter() - t
            ent["all_cores"] = {"value": cores * len(r) / dta, "unit": "samples/s", "cores": cores,
                                "sample": f"{cores} independent copies of the stream at once, one per process, {dta:.2f} s"}
        out[key] = ent
    return out


def final_system_test_leg(reps=20, warm=3, cpu=True):
    """The reference's own end-to-end case -- `Final System Test.ipynb` cells 5-8: the 8-bit over-the-air recording
    gr5ch1_signal.wav through receiver(mode="A2").receive() -- through the drop-in class, HOST array in, decoded bits on
    the host out (upload, chirp sync with the reference's rule, 3 packets x (20 + 180 + 20) symbols, XOR decode, copies
    back; wall clock, median of `reps` calls after `warm`).  The recording travels as a test fixture (tests/golden)."""
    import contextlib, hashlib, io
    path = os.path.join(ROOT, "tests", "golden", "g6_realrec.npz")
    if not os.path.exists(path):
        return {}
    g = np.load(path)
    wav = g["wav_u8"]
    from gf3_audio_modem_amd.OFDM import receiver
    times = []
    with contextlib.redirect_stdout(io.StringIO()):                  # (the class prints the reference's banners)
        rx = receiver(mode="A2", encoding="XOR")
        for i in range(warm + reps):
            t = time.perf_counter()
            bits, _, _ = rx.receive(wav)
            dt = time.perf_counter() - t
            if i >= warm:
                times.append(dt)
    med = float(np.median(times))
    src = np.unpackbits(g["src_bits"])[: int(g["n_src"])]
    ber = float(np.sum(bits[: len(src)] != src) / len(src))
    out = {"workload": f"Final System Test.ipynb: gr5ch1_signal.wav ({len(wav)} samples, 8-bit PCM, 48 kHz, over the air), mode A2 "
                       f"(N=4096, CP=224, bins 100-1499, QPSK, XOR), receiver.receive() from a host array to bits on the host",
           "ms_per_call": med * 1e3, "samples_per_s": len(wav) / med, "timing": f"wall clock, median of {reps} calls after {warm}",
           "bits": int(len(bits)), "bits_sha256_equal_reference": hashlib.sha256(bits.astype(np.uint8).tobytes()).hexdigest() == str(g["sha256_bits"]),
           "ber_vs_source": repr(ber), "ber_string_equal_reference": repr(ber) == str(g["ber_str"]),
           "reference_as_written_s": 20.3, "reference_note": "OFDM.py unmodified, 1 thread, measured in the build container (BASELINE.md section 2), not on this host"}
    # the demodulation stage alone, HIP events on the launch stream: one packet per workgroup (3 workgroups x 220 symbols)
    # against the two-phase form the library picks for this geometry (pilot sums, estimate, data symbols over the chip)
    eng = rx._engine(wav.dtype)
    x = eng._samples(wav)
    starts = (eng.sync_stream(x) + 2)[:-1]
    want = ("Hs", "He", "slope")
    o1, o2 = eng.demod_frames(x, starts, want=want, split=False), eng.demod_frames(x, starts, want=want)
    out["demod_frames"] = {"plan": eng.demod_plan(int(starts.numel())),
                           "one_launch_ms": _event_ms(lambda: eng.demod_frames(x, starts, want=want, split=False), reps, warm),
                           "as_dispatched_ms": _event_ms(lambda: eng.demod_frames(x, starts, want=want), reps, warm),
                           "bits_equal": bool(torch.equal(o1["bits"], o2["bits"])),
                           "Hs_He_slope_bit_identical": all(bool(torch.equal(o1[k], o2[k])) for k in want),
                           "timing": f"HIP events, median of {reps} launches after {warm}"}
    if cpu:
        from oracle import gf3_oracle as orc                         # the CPU baseline of this leg: the restatement on this host
        try:
            from threadpoolctl import threadpool_limits
        except Exception:
            threadpool_limits = None
        pts, bt = orc.qpsk_table()
        p = orc.RxParams(N=4096, CP=224, P=20, D=180, lo=100, hi=1500, const_points=pts, const_bits=bt, known_bits=g["known_bits"])
        ctx = threadpool_limits(limits=1) if threadpool_limits else None
        try:
            r64 = wav / 1.0
            orc.xor_decode(orc.receive(r64, p)["bits"], p)                                    # (warm-up)
            t = time.perf_counter(); cb = orc.xor_decode(orc.receive(r64, p)["bits"], p); dt = time.perf_counter() - t
        finally:
            if ctx is not None and hasattr(ctx, "unregister"):
                ctx.unregister()
        out["cpu_baseline"] = {"value": len(wav) / dt, "unit": "samples/s", "cores": 1, "kind": "port",
                               "sample": f"the whole recording through oracle.receive + xor_decode once, {dt:.3f} s; bits equal the GPU's: {bool(np.array_equal(cb, bits))}"}
    return {"final_system_test": out}


def pcm16_leg(args, cfg, big, payload, steps=20, warm=3):
    """The headline step on the SAME packets stored as 16-bit PCM (the format recordings arrive in) instead of
    f32: half the sample bytes, the same arithmetic.  A secondary figure -- `value` stays the f32-stored one SURVEY 8(d)
    prescribes -- and a test of the energy budget of DESIGN 8.0, which prices the bytes that no longer move."""
    import dataclasses
    from gf3_audio_modem_amd import Engine
    F = big.shape[0]
    scale = 20000.0 / float(big.abs().max())
    big16 = torch.empty(big.shape, dtype=torch.int16, device=big.device)
    rows = 4096
    for i in range(0, F, rows):                                   # (piecewise: no second f32 copy of the batch)
        big16[i:i + rows] = torch.round(big[i:i + rows] * scale).to(torch.int16)
    eng16 = Engine(dataclasses.replace(cfg, in_dtype=torch.int16))
    bits = torch.empty((F, eng16.bytes_per_frame), dtype=torch.uint8, device=big.device)
    starts = torch.empty((F,), dtype=torch.int64, device=big.device)

    def step():
        eng16.sync_frames(big16, F, args.stride, WIN_LO, WIN_LO + args.window, out_starts=starts)
        eng16.demod_frames(big16, starts, out_bits=bits)
    for _ in range(warm):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(steps):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    errs = int((bits != payload).sum().item())                    # (bytes that differ: 0 means every bit is right)
    n_samples = F * args.stride
    del big16, eng16
    return {"pcm16_storage": {"workload": f"the headline step on the same {F} packets stored as int16 PCM (scaled to +-20000) instead of f32",
                              "ms_per_step": ms, "value": n_samples / (ms * 1e-3), "unit": "samples/s", "steps": steps,
                              "payload_bytes_wrong": errs, "sample_bytes_per_step": 2 * n_samples}}


def screened_sync_leg(args, eng, big, payload, exp_starts, steps=20, warm=3):
    """The headline step with the OPT-IN screened frames sync in place of the all-fp64 one (gf3_sync_frames_ex mode 1,
    gf3rx_fscreen.h): every search window in fp32 with a proven bound, the decision taken where the bound decides it, the
    all-fp64 kernel on the windows it does not (none on this clean batch).  The indices are the fp64 kernel's by
    construction and are checked here against the expected offsets; the demodulation is the same all-fp64 kernel.  A
    secondary figure: `value` stays the all-fp64 step."""
    F = big.shape[0]
    starts = torch.empty((F,), dtype=torch.int64, device=big.device)
    bits = torch.empty((F, eng.bytes_per_frame), dtype=torch.uint8, device=big.device)
    work = eng.sync_frames_workspace(F)

    def sync():
        eng.sync_frames(big, F, args.stride, WIN_LO, WIN_LO + args.window, out_starts=starts, screened=True, work=work)

    def step():
        sync()
        eng.demod_frames(big, starts, out_bits=bits)
    ms_sync = _event_ms(sync, steps, warm)
    for _ in range(warm):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(steps):
        step()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    n_samples = F * args.stride
    return {"screened_sync": {"workload": "the headline step with gf3_sync_frames_ex mode 1 (fp32 screen with a proven bound per window + fp64 kernel on "
                                          "unresolved windows) in front of the same all-fp64 demodulation",
                              "dtype": "f32 screen + f64 decisions", "ms_per_step": ms, "value": n_samples / (ms * 1e-3), "unit": "samples/s",
                              "steps": steps, "sync_ms": ms_sync, "sync_timing": f"median of {steps} launches after {warm} (HIP events)",
                              "windows_sent_to_fp64": int(work[:4].view(torch.int32).item()), "windows": F,
                              "sync_exact": bool(torch.equal(starts, exp_starts)), "payload_bytes_wrong": int((bits != payload).sum().item()),
                              "note": "a secondary figure: `value` and `roofline_sync` are the all-fp64 sync"}}


def _event_ms(fn, reps=20, warm=3):
    """SURVEY 8(d) protocol for the secondary legs: MEDIAN HIP-event time of `fn` (one launch on the current stream)
    over `reps` >= 20 launches after `warm` = 3 untimed ones, in ms"""
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def measured_copy_bandwidth(dev, gib=4, reps=20):
    """"also report against the measured copy-kernel bandwidth" -- a device-to-device copy of `gib` GiB
    (far past the Infinity Cache), bytes read + bytes written per second, median of 20 after 3 warm-ups."""
    n = gib << 28                                           # float32 elements
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    ms = _event_ms(lambda: dst.copy_(src), reps)
    del src, dst
    return {"GB_per_s": 2 * n * 4 / ms / 1e6, "what": f"torch device-to-device copy of {gib} GiB: bytes read + written per second, median of {reps} after 3 warm-ups"}


def add_copy_fraction(obj, copy_gbs):
    """frac_of_measured_copy next to every `frac` of an HBM-bound roofline entry (nested dicts included)"""
    if isinstance(obj, dict):
        if obj.get("bound") == "hbm" and obj.get("achieved") is not None and obj.get("unit") == "GB/s":
            obj["frac_of_measured_copy"] = obj["achieved"] / copy_gbs
        for v in list(obj.values()):
            add_copy_fraction(v, copy_gbs)


def config5_rooflines(dev, log2_samples=28, log2_symbols=26, reps=20):
    """BASELINE config 5, sized past the 256 MB Infinity Cache so that the figure is an HBM figure: per N in
    {1024, 2048, 4096, 8192} ONE launch of rfft_kernel over 2^28 f32 samples (1 GiB in, 2.1 GB of complex128 bins
    out, > 0.3 ms), and one launch of the 64-QAM soft demapper over 2^26 symbols (1 GiB in, 1.6 GB of f32 LLRs out).
    Algorithmic bytes per SURVEY 8(d): B_in N + 16 (N/2+1) per transform; 16 + 4 mu per symbol."""
    from gf3_audio_modem_amd import Engine, RxConfig, qpsk_table, square_qam_table
    total = 1 << log2_samples
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(total, dtype=torch.float32, device=dev, generator=gen)
    out = {"roofline_rfft": {}}
    pts, bt = qpsk_table()
    for N in (1024, 2048, 4096, 8192):
        K = N // 2 - 1
        eng = Engine(RxConfig(N=N, CP=0, P=1, D=1, data_bins=np.arange(1, K), const_points=pts, const_bits=bt,
                              known_bits=np.zeros(2 * K, np.uint8), in_dtype=torch.float32, fit_lo=10, fit_hi=100))
        n_sym = total // N
        off = torch.arange(n_sym, dtype=torch.int64, device=dev) * N
        X = torch.empty((n_sym, N // 2 + 1), dtype=torch.complex128, device=dev)
        ms = _event_ms(lambda: eng.rfft_batch(x, off, out=X), reps)
        rows = [0, n_sym // 2 + 1, n_sym - 1]
        ref = np.fft.rfft(np.stack([x[r * N:(r + 1) * N].cpu().numpy().astype(np.float64) for r in rows]))
        err = float(np.abs(X[rows].cpu().numpy() - ref).max() / np.abs(ref).max())
        by = n_sym * (4 * N + 16 * (N // 2 + 1))
        out["roofline_rfft"][f"N{N}"] = {"kernel": f"rfft_kernel<{N // 2},f32>", "bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by,
                                         "avg_launch_ms": ms, "timing": "median of 20 launches after 3 warm-ups", "transforms_per_launch": n_sym, "max_rel_err_vs_numpy_fft": err}
        del X, off
        eng.close()
    del x
    n = 1 << log2_symbols
    pts6, bt6 = square_qam_table(6)
    eng = Engine(RxConfig(N=1024, CP=0, P=1, D=1, data_bins=np.arange(1, 511), const_points=pts6, const_bits=bt6,
                          known_bits=np.zeros(511 * 6, np.uint8), in_dtype=torch.float32, fit_lo=10, fit_hi=100))
    idx = torch.randint(0, 64, (n,), device=dev, generator=gen)
    sym = torch.as_tensor(pts6, device=dev)[idx] + 0.08 * torch.complex(torch.randn(n, dtype=torch.float64, device=dev, generator=gen),
                                                                      torch.randn(n, dtype=torch.float64, device=dev, generator=gen))
    llr = torch.empty((n, 6), dtype=torch.float32, device=dev)
    ms = _event_ms(lambda: eng.soft_demap(sym, 0.0128, out=llr), reps)
    hard, _ = eng.demap_hard(sym[: 1 << 20])
    sign_ok = bool(torch.equal((llr[: 1 << 20] < 0).to(torch.uint8), hard))
    by = n * (16 + 4 * 6)
    out["roofline_soft_demap"] = {"kernel": "soft_demap_bin_kernel<3,3> (64-QAM)", "bound": "hbm", "achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by,
                                  "avg_launch_ms": ms, "symbols_per_launch": n, "sign_equals_hard_decision": sign_ok}
    eng.close()
    return out


def demod_16qam_roofline(dev, args, reps=20):
    """The config-2 geometry with the 16-QAM Gray table: the bits-only table mode of the fused kernel
    (demod_kernel<2048,f32,MODE_SCAN>) over F distinct frame buffers, every bit checked.
    Algorithmic bytes: B_in M N + D C mu / 8 = 204 792 B per packet."""
    from gf3_audio_modem_amd import Engine, RxConfig, square_qam_table
    N, CP, P, D = 4096, 512, 2, 8
    K = N // 2 - 1
    pts, bt = square_qam_table(4)
    known = np.unpackbits(np.load(os.path.join(ROOT, "gf3_audio_modem_amd", "data", "known_bits.npz"))["packed"])
    known = np.tile(known, -(-K * 4 // len(known)))
    cfg = RxConfig(N=N, CP=CP, P=P, D=D, data_bins=np.arange(1, K), const_points=pts, const_bits=bt, known_bits=known,
                   in_dtype=torch.float32, max_window=args.window)
    eng = Engine(cfg)
    F = args.frames
    gen = torch.Generator(device=dev).manual_seed(16)
    payload = torch.randint(0, 256, (F, eng.bytes_per_frame), dtype=torch.uint8, device=dev, generator=gen)
    filler = np.zeros(K, dtype=complex)
    filler[K - 1] = pts[5]
    rows = eng.tx_frames(payload, filler, stride=args.stride, out_dtype=torch.float32)
    starts = torch.arange(F, device=dev, dtype=torch.int64) * args.stride + cfg.chirp_length
    bits = torch.empty_like(payload)
    ms = _event_ms(lambda: eng.demod_frames(rows, starts, out_bits=bits), reps)
    by = F * (4 * cfg.M * N + eng.bytes_per_frame)
    ok = bool(torch.equal(bits, payload))
    eng.close()
    return {"roofline_demod_16qam": {"kernel": "demod_kernel<2048,f32,MODE_SCAN> (16-QAM, bits only)", "bound": "hbm", "achieved": by / ms / 1e6,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS,
                                     "algorithmic_bytes_per_launch": by, "avg_launch_ms": ms, "payload_recovered": ok}}


VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 4      # 1 024 SIMDs, one wave64 VALU instruction per 4 cycles, 2.4 GHz


def stream_sync_roofline(dev, frames=4096, pmc=None, h2d=True):
    """BASELINE config 3 (tools/config3.py): 4 096 16-QAM packets as ONE stream through the measured 30-tap channel,
    stream-mode chirp sync with the reference's global-max / first-extremum / suppression rule, then demod; median of
    20 passes after 3 warm-ups.  The call is NOT memory-bound: the fp32 screen transforms every sample
    twice, so the line is priced against fp32 VALU ISSUE (wave-instructions of one call, from the committed SQ
    counter pass, x 4 cycles / 1 024 SIMDs / 2.4 GHz) and carries the HBM figure beside it -- B_in per sample in,
    8 B per detected peak out -- and the time of the all-fp64 evaluation of the same convolution (mode 1)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gf3_config3", os.path.join(ROOT, "tools", "config3.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    eng, cfg, channel = tool.make_engine()
    r, payload = tool.make_stream(eng, channel, frames)
    res, starts, _ = tool.measure(eng, cfg, r, payload, reps=20, warm=3, fp64_reps=20)
    by = 4 * r.numel() + 8 * (frames + 1)
    t = res["sync_stream_s"]
    insts = (pmc or {}).get("stream_sync_valu_wave_instructions_per_call")
    traffic = (pmc or {}).get("stream_sync_hbm_bytes_per_call")
    out = {"roofline_stream_sync": {"kernel": "gf3_sync_stream (config 3: one 321 M-sample stream, 4 096 packets; scr_ring_kernel + scr_refine_kernel + lists)",
                                    "bound": "valu_f32",
                                    "achieved": (insts / t / 1e9) if insts else None, "peak": VALU_PEAK_WAVE_INSTR_PER_S / 1e9,
                                    "unit": "G wave-instr/s", "frac": (insts / t / VALU_PEAK_WAVE_INSTR_PER_S) if insts else None,
                                    "valu_wave_instructions_per_call": insts,
                                    "counter_source": "profiles/traffic_current.json (SQ_INSTS_VALU pass of this command; not measured in this run)" if insts else None,
                                    "hbm": {"bound": "hbm", "achieved": by / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / t / 1e9 / HBM_PEAK_GBS,
                                            "algorithmic_bytes_per_call": by, "traffic": traffic},
                                    "call_ms": t * 1e3, "fp64_path_ms": res["sync_stream_fp64_path_s"] * 1e3, "demod_ms": res["demod_s"] * 1e3,
                                    "timing": res["timing"],
                                    "samples_per_s_sync_plus_demod": res["samples_per_s"], "ber_vs_payload": res["ber"],
                                    "sync_offsets_exact": res["sync_offsets_as_expected_plus1"],
                                    "sync_path": res["sync_path"]}}
    if h2d:
        out.update(host_ingest_leg(eng, r, starts))
    eng.close()
    return out


def host_ingest_leg(eng, r, starts_one_shot, chunk_samples=1 << 25, reps=3):
    """SURVEY 8(d) "H2D upload excluded and reported separately": the same config-3 stream from PINNED HOST memory
    through Engine.receive_host -- 32 Mi-sample pieces, double-buffered H2D on a copy stream under the previous piece's
    kernels, the reference's global-max rule kept exact across the pieces (gf3_sync_chunk / gf3_sync_decide).  Never
    `value`: the headline is quoted with the samples resident in HBM."""
    host = torch.empty(r.numel(), dtype=r.dtype).pin_memory()
    host.copy_(r)
    torch.cuda.synchronize()
    dst = torch.empty_like(r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(host, non_blocking=True); torch.cuda.synchronize()
    e0.record(); dst.copy_(host, non_blocking=True); e1.record(); torch.cuda.synchronize()
    copy_only = host.numel() * host.element_size() / (e0.elapsed_time(e1) * 1e-3)
    del dst
    ts, res = [], None
    for _ in range(1 + reps):                                   # one warm-up (pinned staging, workspaces)
        t = time.perf_counter(); res = eng.receive_host(host, chunk_samples=chunk_samples); ts.append(time.perf_counter() - t)
    t = float(np.median(ts[1:]))
    one = eng.demod_frames(r, starts_one_shot)["bits"]
    info = res["info"]
    # ... and from PAGEABLE memory, the kind of array the reference's entry point is handed (wavfile.read): the default path
    pageable = host.numpy().copy()
    tp, resp = [], None
    for _ in range(1 + reps):
        t0 = time.perf_counter(); resp = eng.receive_host(pageable); tp.append(time.perf_counter() - t0)
    tpm = float(np.median(tp[1:]))
    pg = {"source": resp["info"]["source"], "chunks": resp["info"]["chunks"], "seconds": tpm, "samples_per_s_with_upload": r.numel() / tpm,
          "h2d_GB_per_s_achieved": resp["info"]["h2d_bytes"] / tpm / 1e9,
          "peaks_equal_one_shot": bool(torch.equal(resp["peaks"], res["peaks"])), "bits_equal_one_shot": bool(torch.equal(resp["bits"], one))}
    del pageable, resp
    return {"h2d": {"from_pageable_memory": pg, "path": "Engine.receive_host: pinned host array -> two device buffers, H2D on a copy stream under the previous piece's "
                            "kernels; all-fp64 chunked sync with the exact global-max rule + demod per piece",
                    "workload": "BASELINE config 3 stream (321 M f32 samples, 4 096 16-QAM packets) from pinned host memory",
                    "samples": int(r.numel()), "chunks": info["chunks"], "chunk_samples": info["chunk_samples"],
                    "seconds": t, "timing": f"median of {reps} passes after 1 warm-up (host wall clock, synchronised)",
                    "samples_per_s_with_upload": r.numel() / t, "h2d_GB_per_s_achieved": info["h2d_bytes"] / t / 1e9,
                    "h2d_copy_only_GB_per_s": copy_only / 1e9,
                    "peaks_equal_one_shot": bool(torch.equal((res["peaks"] + 2)[:-1], starts_one_shot)),
                    "bits_equal_one_shot": bool(torch.equal(res["bits"], one)),
                    "second_look_chunks": info["second_look_chunks"], "second_look_packets": info["second_look_packets"],
                    "seconds_setup_pieces_final": [info["setup_seconds"], info["pieces_seconds"], info["seconds"] - info["setup_seconds"] - info["pieces_seconds"]],
                    "note": "reported separately, never `value` (SURVEY 8d)"}}


def rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def launch_ranks(n):
    """`python bench.py --gpus N` outside torchrun: the parent, which has not touched the GPU, starts N fresh
    copies of this script as child processes (one per rank, torchrun's environment, rendezvous on 127.0.0.1),
    relays rank 0's JSON line, and returns a non-zero code if any rank failed.  No os.exec*."""
    from gf3_audio_modem_amd.dist import spawn_ranks
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if os.environ.get("GF3_BENCH_RANK_CMD"):             # tests: a stub in place of the rank program
        cmd = json.loads(os.environ["GF3_BENCH_RANK_CMD"]) + sys.argv[1:]
    rc, _ = spawn_ranks(cmd, n)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per GPU (default: 65 536 at N=1 = BASELINE config 2; 131 072 at N>1 = config 4)")
    ap.add_argument("--stride", type=int, default=78720)
    ap.add_argument("--window", type=int, default=320)
    ap.add_argument("--chunks", type=int, default=8, help="N>1: pieces the batch is cut into to overlap the all-gather")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the 2 s sustained-power probe after the timed region")
    ap.add_argument("--rehearse-chunked", action="store_true",
                    help="N=1 only: run the N>1 code path (chunked launches, per-chunk all-gather on a side stream) "
                         "through a one-rank RCCL group; a rehearsal of the multi-GPU path, not the headline number")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 rFFT / soft-demap roofline legs (N=1)")
    ap.add_argument("--no-by-n", action="store_true", help="skip the legs that price the fused kernels at N = 1024, 2048, 8192 (N=1)")
    ap.add_argument("--no-stream", action="store_true", help="skip the config-3 stream-sync roofline leg (N=1)")
    ap.add_argument("--no-screened-sync", action="store_true", help="skip the leg that repeats the step with the opt-in screened frames sync (N=1)")
    ap.add_argument("--no-pcm16", action="store_true", help="skip the leg that repeats the step on int16-stored samples (N=1)")
    ap.add_argument("--no-final-system-test", action="store_true", help="skip the leg that runs the reference's own recording through the drop-in class (N=1)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-ingest (pinned, chunked H2D + stream receive) leg (N=1)")
    ap.add_argument("--gather-algo", default=None, help="N>1: NCCL_ALGO for the all-gather (e.g. Ring, Tree, Direct), set before the "
                    "process group exists, so that a scaling run can compare algorithms (SURVEY section 5) without a code change")
    ap.add_argument("--gather-proto", default=None, help="N>1: NCCL_PROTO (e.g. Simple, LL, LL128)")
    args = ap.parse_args()
    if args.gather_algo:
        os.environ["NCCL_ALGO"] = args.gather_algo         # (inherited by the ranks launch_ranks starts)
    if args.gather_proto:
        os.environ["NCCL_PROTO"] = args.gather_proto
    # RCCL's own stream at high priority, so that the per-chunk all-gathers never share a hardware queue with the compute
    # streams they are meant to run under (recorded with the collective's other settings)
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))        # the children run the rest of main() with WORLD_SIZE set
    if args.frames <= 0:
        args.frames = 65536 if args.gpus == 1 else 131072

    pool = cores = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu and not args.rehearse_chunked:
        pool, cores = make_cpu_pool()                     # before the GPU is initialised (see make_cpu_pool)
    from gf3_audio_modem_amd import dist as gd
    rank, world, local = gd.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1
    if args.rehearse_chunked and world == 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        multi = True

    eng, cfg, big, payload, gaps = build_workload(args, rank)
    F = args.frames
    n_samples = F * args.stride
    exp_starts = torch.arange(F, device=dev, dtype=torch.int64) * args.stride + gaps + cfg.chirp_length

    class Run:
        """One way of running the step: N=1 (one launch of each kernel over the whole batch) or N>1 with the batch
        cut into `chunks` pieces whose packed bits are all-gathered on a side stream as soon as they exist
        (chunks = 1: the literal single all-gather after the kernels)."""

        def __init__(self, chunks, gather=True):
            self.chunks = chunks if (multi and F % chunks == 0) else 1
            self.Fc = F // self.chunks
            self.bits = torch.empty((F, eng.bytes_per_frame), dtype=torch.uint8, device=dev)
            # gather=False: the same launches with the collective REMOVED (what the exchange step costs is the difference)
            self.gathered = torch.empty((F * world, eng.bytes_per_frame), dtype=torch.uint8, device=dev) if (multi and gather) else None
            self.og = gd.OverlappedGather(self.gathered, F, self.chunks) if (multi and gather) else None
            self.starts_all = torch.empty((F,), dtype=torch.int64, device=dev)
            self.starts_c = torch.empty((self.chunks, self.Fc), dtype=torch.int64, device=dev)   # per-chunk sync results (chunk-relative)
            # (a HIGH-PRIORITY stream: streams of one priority level can share a hardware queue, which would serialise the sync
            #  launches behind the demod launches they are meant to run ahead of -- DESIGN 3.2 has the measurement that showed it)
            self.s_sync = torch.cuda.Stream(priority=-1) if multi else None   # chunked path: sync kernels run ahead on their own stream
            self.s_dem2 = torch.cuda.Stream() if multi else None      # ... and odd chunks' demod kernels on a second one
            self.ev_sync = [torch.cuda.Event() for _ in range(self.chunks)]
            self.ev_step = torch.cuda.Event()
            self.chunk_base = (torch.arange(self.chunks, device=dev, dtype=torch.int64) * (self.Fc * args.stride))[:, None]

        def step(self, ev=None):
            if not multi:
                if ev: ev[0].record()
                starts = eng.sync_frames(big, F, args.stride, WIN_LO, WIN_LO + args.window, out_starts=self.starts_all)
                if ev: ev[1].record()
                eng.demod_frames(big, starts, out_bits=self.bits)
                if ev: ev[2].record()
                return starts
            chunks, Fc = self.chunks, self.Fc
            # the sync launches of all chunks go to their own stream and run ahead; demod of chunk c waits for its
            # sync only, so the two kernel families overlap and a chunk boundary costs no drained-GPU tail
            main = torch.cuda.current_stream()
            self.ev_step.record(main)
            self.s_sync.wait_event(self.ev_step)                   # previous step's demods are done with starts_c
            with torch.cuda.stream(self.s_sync):
                for c in range(chunks):
                    if ev and c == 0: ev[0].record()
                    eng.sync_frames(big[c * Fc:(c + 1) * Fc], Fc, args.stride, WIN_LO, WIN_LO + args.window, out_starts=self.starts_c[c])
                    if ev and c == 0: ev[1].record()
                    self.ev_sync[c].record()
            self.s_dem2.wait_event(self.ev_step)
            for c in range(chunks):                           # demod of consecutive chunks on alternating streams: the tail
                with torch.cuda.stream(main if c % 2 == 0 else self.s_dem2):   # of one launch overlaps the head of the next
                    torch.cuda.current_stream().wait_event(self.ev_sync[c])
                    if ev and c == 0: ev[3].record()
                    eng.demod_frames(big[c * Fc:(c + 1) * Fc], self.starts_c[c], out_bits=self.bits[c * Fc:(c + 1) * Fc])
                    if ev and c == 0: ev[2].record()
                    if self.og is not None:
                        self.og.chunk_done(c, self.bits[c * Fc:(c + 1) * Fc])
            main.wait_stream(self.s_dem2)
            main.wait_stream(self.s_sync)
            torch.add(self.starts_c, self.chunk_base, out=self.starts_all.view(chunks, Fc))
            if self.og is not None:
                self.og.finish()
            return self.starts_all

        def timed(self, warmup, steps):
            """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides;
            returns (seconds = max over ranks, per-step event sets); self.t_local = this rank's own seconds, taken
            before the closing barrier."""
            for _ in range(warmup):
                self.step()
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
            gd.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                self.step(evs[k])
            torch.cuda.synchronize()
            self.t_local = time.perf_counter() - t0
            gd.barrier()
            return gd.max_over_ranks(time.perf_counter() - t0, dev), evs

        def verify(self):
            """correctness of what was just timed: EVERY frame decodes to its payload, every sync offset exact,
            and (N>1) every rank holds every rank's bits in the documented block-cyclic order"""
            x = torch.bitwise_xor(self.bits, payload)
            bit_errors = int(torch.count_nonzero(x).item())      # differing bytes (0 => BER 0)
            if bit_errors:
                bit_errors = int(np.unpackbits(x.cpu().numpy()).sum())
            sync_ok = bool(torch.equal(self.starts_all, exp_starts))
            gather_ok = None
            if multi and self.og is not None:
                import torch.distributed as tdist
                mine = gd.cyclic_frame_index(rank, world, F, self.chunks).to(dev)
                gather_ok = bool(torch.equal(self.gathered[mine], self.bits))
                # ... and this rank's copy of every OTHER rank's rows matches what that rank decoded (a checksum of
                # each rank's rows, weighted by row number, is exchanged and recomputed on the local copy)
                w = (torch.arange(F, device=dev, dtype=torch.int64) % 65521 + 1)[:, None]

                def checksum(rows):
                    return (rows.to(torch.int64) * w).sum()
                sums = torch.zeros(world, dtype=torch.int64, device=dev)
                sums[rank] = checksum(self.bits)
                tdist.all_reduce(sums, op=tdist.ReduceOp.SUM)
                for r in range(world):
                    idx = gd.cyclic_frame_index(r, world, F, self.chunks).to(dev)
                    gather_ok = gather_ok and bool(checksum(self.gathered[idx]) == sums[r])
                flag = torch.tensor([1.0 if gather_ok else 0.0, float(sync_ok)], dtype=torch.float64, device=dev)
                tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)           # every rank's verdict, not rank 0's alone
                gather_ok = bool(flag[0].item() == 1.0)
                sync_ok = bool(flag[1].item() == 1.0)
                be = torch.tensor([bit_errors], dtype=torch.int64, device=dev)
                tdist.all_reduce(be, op=tdist.ReduceOp.SUM)             # bit errors of ALL ranks' frames, each checked by its own rank
                bit_errors = int(be.item())
            return bit_errors, sync_ok, gather_ok

    run = Run(args.chunks)
    dt, evs = run.timed(args.warmup, args.steps)
    bit_errors, sync_ok, gather_ok = run.verify()
    chunks, Fc = run.chunks, run.Fc
    step = run.step

    # sustained package power: the same step repeated for ~2 s after the timed region (the hwmon sensor averages
    # over a window far longer than a 20-step run), median of the second half of the samples
    def sustained_probe(r, seconds, ms_step):
        """`r.step` repeated for ~`seconds` inside the same bracket as the timed region (barrier + synchronize on both
        sides, maximum over ranks), rank 0 sampling its package power: (record, watts or None)"""
        n_probe = int(seconds / (ms_step * 1e-3)) + 1           # ms_step is the max over ranks: the same count everywhere
        power = PowerSampler(local) if rank == 0 else None
        if power is not None:
            power.__enter__()
        gd.barrier()
        torch.cuda.synchronize()
        t_probe = time.perf_counter()
        for _ in range(n_probe):                               # every rank steps (the N>1 step holds a collective)
            r.step()
        torch.cuda.synchronize()
        gd.barrier()
        t_probe = gd.max_over_ranks(time.perf_counter() - t_probe, dev)
        rec = {"steps": n_probe, "seconds": t_probe, "ms_per_step": t_probe / n_probe * 1e3,
               "value": world * n_samples * n_probe / t_probe, "unit": "samples/s"}
        w = None
        if power is not None:
            power.__exit__()
            if power.samples:
                w = float(np.median(power.samples[len(power.samples) // 2:]))
        gd.barrier()
        return rec, w

    power_w, sustained = None, None
    if not args.no_power:
        # the same step over a window a hundred times longer than a 20-step timed region: a second reading of `value`,
        # not a replacement for it
        sustained, power_w = sustained_probe(run, 2.0, dt / args.steps * 1e3)

    # N>1: the same batch once more with the literal single all-gather of the north_star (one collective per step,
    # issued after the kernels, nothing overlapped), and -- so that one scaling run says what the exchange step costs --
    # both forms again with the collective REMOVED.  Every hot kernel alone holds the package at its power cap (DESIGN
    # 8.0), so RCCL's copy kernels under the next chunk's compute are paid for in watts and CUs: each variant carries
    # rank 0's sustained package power beside its time, and the gather's cost is given per rank (max and min over ranks
    # of that rank's own seconds with the collective minus without).
    single = no_gather = gather_cost = None
    if multi:
        t_chunked_local = run.t_local

        def variant(chunks_v, gather):
            rv = Run(chunks_v, gather)
            dtv, _ = rv.timed(min(args.warmup, 2), args.steps)
            bev, sov, gov = rv.verify()
            rec = {"chunks": rv.chunks, "collective": bool(gather), "ms_per_step": dtv / args.steps * 1e3,
                   "value": world * n_samples * args.steps / dtv, "unit": "samples/s", "bit_errors": bev, "sync_exact": sov, "gather_exact": gov}
            if not args.no_power:
                sus, w = sustained_probe(rv, 1.5, dtv / args.steps * 1e3)
                rec.update(sustained_ms_per_step=sus["ms_per_step"], package_power_w_rank0=w)
            t_loc = rv.t_local
            del rv
            return rec, t_loc

        def per_rank_cost(t_with, t_without):
            c = gd.gather_floats((t_with - t_without) / args.steps * 1e3, dev)
            return {"max_over_ranks": max(c), "min_over_ranks": min(c), "per_rank": c}

        chunks_main = run.chunks
        del run.gathered, run.og
        run.og = None
        no_gather, t_ng = variant(chunks_main, False)
        gather_cost = {"chunked_ms": per_rank_cost(t_chunked_local, t_ng),
                       "what": "this rank's seconds per step with the collective minus the same launches without it; "
                               "chunked: per-chunk all-gathers on a side stream under the next chunk's kernels"}
        if chunks_main != 1:
            single, t_single = variant(1, True)
            ng1, t_ng1 = variant(1, False)
            no_gather["single_launch_form"] = ng1
            gather_cost["single_ms"] = per_rank_cost(t_single, t_ng1)
        if power_w is not None:
            gather_cost["package_power_w_rank0"] = {"chunked": power_w, "no_gather": no_gather.get("package_power_w_rank0"),
                                                    "single": (single or {}).get("package_power_w_rank0"),
                                                    "no_gather_single": no_gather.get("single_launch_form", {}).get("package_power_w_rank0")}

    t_sync = float(np.mean([e[0].elapsed_time(e[1]) for e in evs])) * 1e-3
    t_demod = float(np.mean([(e[3] if multi else e[1]).elapsed_time(e[2]) for e in evs])) * 1e-3
    b_in = 4
    Fl = Fc if multi else F                      # frames per timed launch
    bytes_demod = Fl * (b_in * cfg.M * cfg.N + eng.bytes_per_frame)                     # 200 700 B/frame
    bytes_sync = Fl * (b_in * (cfg.chirp_length + args.window - 1) + 8)
    ach = bytes_demod / t_demod / 1e9
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/profile_round.sh ->
    # tools/collect_profiles.py -> profiles/traffic_current.json): a constant read from a file, NOT measured by this run
    # -- and only if it was collected on the sources the loaded library was built from (the file carries their hash)
    from gf3_audio_modem_amd import _lib
    lib_ver, lib_src16 = _lib.build_id()
    traffic, pmc, traffic_note = None, {}, None
    tfile = os.path.join(ROOT, "profiles", "traffic_current.json")
    if os.path.exists(tfile):
        try:
            pmc = json.load(open(tfile))
            if pmc.get("source_sha16") != lib_src16:
                traffic_note = (f"profiles/traffic_current.json was collected on sources {pmc.get('source_sha16')}, the loaded library is "
                                f"{lib_src16}: not quoted (rerun tools/profile_round.sh + tools/collect_profiles.py)")
                pmc = {}
            traffic = pmc.get("demod_kernel_bytes_per_launch_at_F", {}).get(str(Fl))   # same launch size only
        except Exception:
            traffic, pmc = None, {}
    extra = {}
    power_obj = None
    if world == 1 and not multi:
        if not args.no_power:
            st_all = run.starts_all
            power_obj = power_roofline(local, lambda: eng.demod_frames(big, st_all, out_bits=run.bits), t_demod, F, lib_src16)
        del run.bits
        if not args.no_screened_sync:
            extra.update(screened_sync_leg(args, eng, big, payload, exp_starts))
        if not args.no_pcm16:
            extra.update(pcm16_leg(args, cfg, big, payload))
        if not args.no_config5:
            extra.update(demod_16qam_roofline(dev, args))
            extra.update(config5_rooflines(dev))
        if not args.no_by_n:
            extra.update(by_N_rooflines(dev, args))
        if not args.no_stream:
            extra.update(stream_sync_roofline(dev, pmc=pmc, h2d=not args.no_h2d))
        if not args.no_final_system_test:
            extra.update(final_system_test_leg(cpu=not args.no_cpu))
        extra["hbm_copy_measured"] = measured_copy_bandwidth(dev)

    if rank == 0:
        ver, src = lib_ver, lib_src16
        total_frames = F * world
        if multi and world > 1:
            workload = (f"BASELINE config 4: {total_frames} frames of the config-2 geometry (N=4096 CP=512 P=2 D=8 QPSK, "
                        f"chirp-prefixed frame buffers) sharded {F} per GPU, windowed chirp sync + LS pilot equalisation + "
                        f"hard demap, RCCL all-gather of the packed decoded bits" + (" (1 048 576 frames at 8 GPUs)" if F == 131072 else ""))
        else:
            workload = ("BASELINE config 2: N=4096 CP=512 P=2 D=8 QPSK, chirp-prefixed frame buffers, "
                        "windowed chirp sync + LS pilot equalisation + hard demap")
        out = {
            "metric": "demod samples/sec + decoded-bit BER vs OFDM.py, N=4096 QPSK, 1/2/4/8 GPU",
            "value": world * n_samples * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "frames_per_gpu": F, "frames_total": total_frames, "samples_per_frame": args.stride, "sample_storage": "f32",
                       "sync_window_lags": args.window, "parallelism": f"frames sharded over {world} GPU(s), "
                       f"packed bits all-gathered in {chunks} chunk(s) under compute" if multi else "single GPU"},
            "ber": bit_errors / (total_frames * cfg.bits_per_frame), "bit_errors": bit_errors,
            "frames_checked": total_frames, "frames_checked_note": "every rank compares its own frames with their payload; the counts are summed over ranks" if multi else None,
            "sync_exact": sync_ok,
            "gather_exact": gather_ok, "ranks_in_group": (torch.distributed.get_world_size() if multi else 1),
            "single_gather": single, "no_gather": no_gather, "gather_cost": gather_cost,
            "sustained_2s": sustained,
            "collective": ({"backend": torch.distributed.get_backend(), "library_version": rccl_version(),
                            "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"),
                            "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "TORCH_NCCL_"))},
                            "note": "algorithm / protocol are RCCL's own choice unless NCCL_ALGO / NCCL_PROTO are set "
                                    "(--gather-algo / --gather-proto)"} if multi else None),
            "library": {"version": ver, "source_sha16": src},
            "roofline": {"kernel": "demod_kernel<2048,f32,MODE_QPSK>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/traffic_current.json (FETCH_SIZE + WRITE_SIZE passes of this command under rocprofv3, "
                                            "gfx950-corrected; a committed constant collected on these sources, not measured in this run)" if traffic else traffic_note),
                         "bound_note": "`bound: hbm` is the contract's roofline (algorithmic bytes / launch time / 8 TB/s); what sets the launch time "
                                       "is the package power cap: see `power`",
                         "power": power_obj,
                         "algorithmic_bytes_per_launch": bytes_demod, "avg_launch_ms": t_demod * 1e3,
                         "package_power_w_sustained": power_w,
                         "energy_nJ_per_sample": (power_w * dt / args.steps / (world * n_samples) * 1e9 * world) if power_w else None},
            "roofline_sync": {"kernel": "corr_kernel<1024,f32> (15 x 2048-point transforms per packet)", "bound": "hbm", "achieved": bytes_sync / t_sync / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_sync / t_sync / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_launch": bytes_sync, "avg_launch_ms": t_sync * 1e3},
        }
        out.update(extra)
        if "hbm_copy_measured" in out:
            add_copy_fraction(out, out["hbm_copy_measured"]["GB_per_s"])
        if world == 1 and not multi and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(cfg, big, payload, args.window, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            if pool is not None:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(pool, cores, cfg, big, payload, args.window)
            out.update(cpu_baseline_streams(pool, cores))
        print(json.dumps(out))
    if pool is not None:
        pool.close(); pool.join()
    if multi:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
