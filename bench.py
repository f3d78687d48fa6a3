#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X polar decoder (BASELINE.json):
codewords/s (+ info-Gbit/s) of Fast-SSCL decoding, N=1024 K=512 L=8 + CRC-24C, batch 2^20 per GPU,
synthetic AWGN LLRs (Eb/N0 = 2 dB, systematic codewords with valid checksums), inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the
script is one rank; started plainly it launches its N ranks itself as fresh child processes -- before anything in the
parent has touched the GPU -- relays rank 0's JSON line and exits non-zero if any rank does.

A step = one pass of the hot path over the whole resident batch: pcd_decode_f32 on device pointers
(+ for N > 1 the all-gather of the decoded bytes and verdicts over RCCL).  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--log2-batch", type=int, default=None, help="codewords per GPU = 2^this (default: BASELINE's 20)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc-traffic", dest="pmc_traffic", action="store_true", default=None,
                    help="fill roofline.traffic from two rocprofv3 --pmc child passes (default at N = 1 when rocprofv3 "
                         "is on PATH; adds about a minute)")
    ap.add_argument("--no-pmc-traffic", dest="pmc_traffic", action="store_false")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="N > 1 started without a launcher: seconds after which the rank processes are terminated")
    ap.add_argument("--kernel-name", action="store_true",
                    help="print the name of the kernel the decoder of --config launches (host only, no GPU) and exit")
    ap.add_argument("--launch-check", action="store_true",
                    help="every rank reports its RANK / WORLD_SIZE / LOCAL_RANK and exits before touching the GPU "
                         "(test of the N > 1 launcher)")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as fresh child processes (this process has not imported
    torch nor loaded the HIP library yet), wait for them, exit with the first non-zero status.  The ranks run in their
    own sessions: on a failing rank, on SIGTERM / SIGINT to this process and after --launch-timeout seconds every rank's
    process group is terminated, then killed -- no rank is left waiting in a collective with a GPU in its hands."""
    import signal
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []

    def stop_all(sig=signal.SIGTERM):
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        stop_all()
        time.sleep(1.0)
        stop_all(signal.SIGKILL)
        sys.exit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          start_new_session=True))
        deadline = time.monotonic() + args.launch_timeout
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                st = p.poll()
                if st is None:
                    continue
                alive.remove(p)
                if st != 0 and rc == 0:
                    rc = st
                    stop_all()               # a rank failed: the others would wait in a collective for ever
            if alive and time.monotonic() > deadline:
                print(f"[bench] ranks still running after --launch-timeout {args.launch_timeout:.0f} s: terminating them",
                      file=sys.stderr)
                rc = rc or 124
                stop_all()
                time.sleep(2.0)
                stop_all(signal.SIGKILL)
                break
    finally:
        stop_all()
        for q in procs:
            try:
                q.wait(timeout=5)
            except subprocess.TimeoutExpired:
                stop_all(signal.SIGKILL)
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

np = torch = pypolar_amd = pdet = sharding = frozen_bits = None


def heavy_imports():
    """numpy / torch / the product's ctypes module: only in a process that is going to be a rank"""
    global np, torch, pypolar_amd, pdet, sharding, frozen_bits
    import numpy
    import torch as _torch
    sys.path.insert(0, os.path.join(ROOT, "polar-codes_amd"))
    import pypolar_amd as _pp
    from pypolar_amd import detector as _pdet
    from pypolar_amd import sharding as _sh
    from pypolar_amd.construction import frozen_bits as _fb
    np, torch, pypolar_amd, pdet, sharding, frozen_bits = numpy, _torch, _pp, _pdet, _sh, _fb

CONFIGS = {
    # SURVEY section 8(d)
    "cfg2": dict(N=1024, K=512, L=1, dsnr=0.0, crc=0, crc_type="crc", ebn0=2.5, log2_batch=20,
                 name="N=1024 K=512 SC, batch=2^20"),
    "cfg3": dict(N=1024, K=512, L=8, dsnr=0.0, crc=24, crc_type="crcnr", ebn0=2.0, log2_batch=20,
                 name="N=1024 K=512 SCL L=8 + CRC-24C, batch=2^20"),
    # the int8 ("char") decoders: int8 rows = round(8 y) clipped to +-127, 1 byte per LLR from HBM
    "cfg3_i8": dict(N=1024, K=512, L=8, dsnr=0.0, crc=24, crc_type="crcnr", ebn0=2.0, log2_batch=20, dectype="char",
                    name="N=1024 K=512 SclFipChar (int8) L=8 + CRC-24C, batch=2^20"),
    "cfg2_i8": dict(N=1024, K=512, L=1, dsnr=0.0, crc=0, crc_type="crc", ebn0=2.5, log2_batch=20, dectype="char",
                    name="N=1024 K=512 FastSscFipChar (int8) SC, batch=2^20"),
    # the 8-GPU configurations of SURVEY 8(d), runnable per GPU (weak scaling: the batch below is per GPU)
    "cfg4": dict(N=1024, K=176, L=8, dsnr=0.0, crc=11, crc_type="crcnr", ebn0=1.0, log2_batch=19, ctor="5G",
                 name="5G-table N=1024 K=176 SCL L=8 + CRC-11, batch=2^19 per GPU (2^22 over 8)"),
    "cfg5": dict(N=4096, K=2048, L=32, dsnr=0.0, crc=32, crc_type="crc", ebn0=1.5, log2_batch=15,
                 name="N=4096 K=2048 SCL L=32 + CRC-32C, batch=2^15 per GPU (2^18 over 8)"),
    # cells of the reference's own benchmark grid (benchmark/main_benchmark.cpp:405-500) that the round-4 geometries of the
    # lane-per-path kernel serve; not BASELINE configurations -- for profiles (tools/collect_profiles.sh)
    "n512_l8": dict(N=512, K=256, L=8, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0, log2_batch=18, name="N=512 K=256 SCL L=8 + CRC-8, batch=2^18"),
    "n2048_l8": dict(N=2048, K=1024, L=8, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0, log2_batch=17, name="N=2048 K=1024 SCL L=8 + CRC-8, batch=2^17"),
    "n1024_l16": dict(N=1024, K=512, L=16, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0, log2_batch=17, name="N=1024 K=512 SCL L=16 + CRC-8, batch=2^17"),
    "n1024_l32": dict(N=1024, K=512, L=32, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0, log2_batch=16, name="N=1024 K=512 SCL L=32 + CRC-8, batch=2^16"),
    "n512_sc": dict(N=512, K=256, L=1, dsnr=0.0, crc=0, crc_type="crc", ebn0=2.0, log2_batch=20, name="N=512 K=256 SC (Fast-SSC), batch=2^20"),
    "n1024_l4": dict(N=1024, K=512, L=4, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0, log2_batch=18, name="N=1024 K=512 SCL L=4 + CRC-8, batch=2^18"),
    "n1024_l2": dict(N=1024, K=512, L=2, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0, log2_batch=19, name="N=1024 K=512 SCL L=2 + CRC-8, batch=2^19"),
    # secondary workload: rate matching in front of the decoder (PolarCode::Puncturer::depuncture<float>)
    "dep3": dict(N=1024, K=512, L=0, dsnr=0.0, crc=0, crc_type="crc", ebn0=2.0, log2_batch=20, M=864,
                 name="depuncture 864 -> 1024 float LLRs (160 frozen positions removed), batch=2^20"),
    # secondary workload: the encoder side of cfg3
    "enc3": dict(N=1024, K=512, L=0, dsnr=0.0, crc=24, crc_type="crcnr", ebn0=2.0, log2_batch=22,
                 name="N=1024 K=512 + CRC-24C, batch=2^22"),
}


def config_frozen_set(cfg):
    """BB construction, or (cfg 4) the 5G construction (3GPP TS 38.212 polar sequence)"""
    return frozen_bits(cfg["N"], cfg["K"], cfg["dsnr"], cfg.get("ctor", "BB"))


def make_batch(cfg, frozen, B, device, seed):
    """synthetic traffic: random payload -> checksum + systematic polar encoding (the product's own HIP encoder,
    parity-tested against the reference in tests/test_gpu_encoder.py) -> BPSK + AWGN.
    returns (llr float32 [B, N], info bytes uint8 [B, nb])"""
    N, K = cfg["N"], cfg["K"]
    nb = (K + 7) // 8
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    enc = pypolar_amd.PolarEncoder(N, frozen, device=device.index)
    enc.setErrorDetection(cfg["crc"], cfg["crc_type"])
    sigma = 1.0 / np.sqrt(2.0 * 10 ** (cfg["ebn0"] / 10.0) * K / N)
    info = torch.randint(0, 256, (B, nb), generator=g, device=device, dtype=torch.uint8)
    code = enc.encode_batch(info)                  # fills the checksum fields of `info` in place
    llr = torch.empty((B, N), dtype=torch.float32, device=device)
    shifts = torch.arange(7, -1, -1, device=device, dtype=torch.uint8)
    chunk = 1 << 17
    for lo in range(0, B, chunk):
        n = min(chunk, B - lo)
        bits = ((code[lo:lo + n].unsqueeze(-1) >> shifts) & 1).reshape(n, N)
        noise = torch.randn((n, N), generator=g, device=device, dtype=torch.float32)
        llr[lo:lo + n] = (1.0 - 2.0 * bits.float()) + sigma * noise
        del bits, noise
    torch.cuda.synchronize()
    return llr, info


def cpu_baseline_encode(cfg, seconds):
    """the reference's CPU encoder (oracle/_ref) on the GPU box's host cores; else the scalar oracle port"""
    cores = len(os.sched_getaffinity(0))
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        try:
            out = subprocess.run([harness, "enctime", str(cfg["N"]), str(cfg["K"]), "BB", str(cfg["dsnr"]),
                                  str(cfg["crc"]), cfg["crc_type"], str(cores), str(seconds)],
                                 capture_output=True, text=True, timeout=seconds * 6 + 120, check=True)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {"value": r["cw_per_s"], "unit": "codewords/s", "cores": cores, "kind": "reference",
                    "sample": f"256 random information words per thread cycled for {r['seconds']:.1f} s, one reference "
                              f"ButterflyFipPacked per thread ({cores} threads), {r['codewords']} encodes"}
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reference harness unavailable ({e}); timing the oracle port", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    frozen = frozen_bits(cfg["N"], cfg["K"], cfg["dsnr"])
    info = np.random.default_rng(0).integers(0, 256, (cfg["K"] + 7) // 8, dtype=np.uint8)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(64):
            ob.encode(cfg["N"], frozen, cfg["crc"], info)
        n += 64
    el = time.time() - t0
    return {"value": n / el, "unit": "codewords/s", "cores": 1, "kind": "port",
            "sample": f"{n} encodes by the scalar C oracle in {el:.1f} s"}


def bench_depuncture(args, cfg, dev, dev_index, rank, world, barrier_fn, emit):
    """secondary workload: batched depuncturing of float LLRs (pure data movement, HBM-bound)"""
    N, M = cfg["N"], cfg["M"]
    log2b = args.log2_batch if args.log2_batch is not None else cfg["log2_batch"]
    B = 1 << log2b
    frozen = frozen_bits(N, cfg["K"], cfg["dsnr"], "BB")
    punc = pypolar_amd.Puncturer(M, frozen, device=dev_index)
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    x = torch.randn((B, M), generator=g, device=dev, dtype=torch.float32)
    out = torch.empty((B, N), dtype=torch.float32, device=dev)
    for _ in range(args.warmup):
        punc.depuncture_batch(x, out=out)
    barrier_fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:  # launched on torch's current stream, so torch events bracket the kernel
        a.record()
        punc.depuncture_batch(x, out=out)
        b.record()
    barrier_fn()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if rank != 0:
        return
    bytes_per_row = 4 * (M + N)          # punctured row in, parent row out
    achieved = bytes_per_row * B / (k_ms * 1e-3) / 1e9
    res = {
        "metric": f"rows/sec, {cfg['name']}", "value": B * world * args.steps / elapsed, "unit": "rows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"] + (" per GPU" if world > 1 else ""), "parent_block_length": N,
                   "block_length": M, "rows_per_gpu": B},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "gather_rows4_kernel",
                     "kernel_ms_avg": k_ms, "algorithmic_bytes_per_row": bytes_per_row, "rows_per_launch": B},
    }
    if world == 1 and args.pmc_traffic:
        t = pmc_traffic(args.config, log2b, "gather_rows4_kernel")
        if t:
            res["roofline"].update({"traffic": t["bytes"], "traffic_unit": "bytes per launch at the L2<->fabric "
                                    "interface (Infinity-Cache hits included): 2 x FETCH_SIZE + WRITE_SIZE",
                                    "traffic_over_algorithmic": t["bytes"] / (bytes_per_row * B)})
    if world == 1 and not args.no_cpu_baseline:
        cores = len(os.sched_getaffinity(0))
        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
        try:
            o = subprocess.run([harness, "punctime", str(M), str(N), str(cfg["K"]), "BB", str(cfg["dsnr"]), str(cores),
                                str(min(args.cpu_seconds, 8.0))], capture_output=True, text=True, timeout=120, check=True)
            r = json.loads(o.stdout.strip().splitlines()[-1])
            res["cpu_baseline"] = {"value": r["rows_per_s"], "unit": "rows/s", "cores": cores, "kind": "reference",
                                   "sample": f"256 rows per thread cycled for {r['seconds']:.1f} s through the reference's "
                                             f"Puncturer::depuncture<float> ({cores} threads), {r['rows']} rows"}
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reference harness unavailable ({e})", file=sys.stderr)
    emit(res)


def bench_encode(args, cfg, dev, dev_index, rank, world, barrier_fn, emit):
    """secondary workload: batched systematic encoding + checksum generation"""
    N, K = cfg["N"], cfg["K"]
    nb = (K + 7) // 8
    log2b = args.log2_batch if args.log2_batch is not None else cfg["log2_batch"]
    B = 1 << log2b
    frozen = frozen_bits(N, K, cfg["dsnr"], "BB")
    enc = pypolar_amd.PolarEncoder(N, frozen, device=dev_index)
    enc.setErrorDetection(cfg["crc"], cfg["crc_type"])
    g = torch.Generator(device=dev)
    g.manual_seed(77 + rank)
    info = torch.randint(0, 256, (B, nb), generator=g, device=dev, dtype=torch.uint8)
    code = torch.empty((B, N // 8), dtype=torch.uint8, device=dev)
    for _ in range(args.warmup):
        enc.encode_batch(info, out=code)
    barrier_fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:  # the encoder is launched on torch's current stream, so torch events bracket the kernel
        a.record()
        enc.encode_batch(info, out=code)
        b.record()
    barrier_fn()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if rank != 0:
        return
    fb = pdet.field_bytes(cfg["crc"])
    bytes_per_cw = nb + N // 8 + fb      # information word in, code word out, checksum field written back
    achieved = bytes_per_cw * B / (k_ms * 1e-3) / 1e9
    cw_s = B * world * args.steps / elapsed
    res = {
        "metric": f"codewords/sec, systematic encode, {cfg['name']}",
        "value": cw_s, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "encode " + cfg["name"] + (" per GPU" if world > 1 else ""), "block_length": N,
                   "info_bits": K, "detector": f"{cfg['crc_type']}-{cfg['crc']}", "codewords_per_gpu": B,
                   "frozen_set": f"BB dSNR {cfg['dsnr']} dB",
                   "parallelism": f"rows sharded over {world} GPU(s), no exchange" if world > 1 else "single GPU"},
        "code_gbit_per_s": cw_s * N / 1e9,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "encode_kernel",
                     "kernel_ms_avg": k_ms, "algorithmic_bytes_per_codeword": bytes_per_cw, "codewords_per_launch": B},
    }
    if world == 1 and args.pmc_traffic:
        t = pmc_traffic(args.config, log2b, "encode_kernel")
        if t:
            res["roofline"].update({"traffic": t["bytes"], "traffic_unit": "bytes per launch at the L2<->fabric "
                                    "interface (Infinity-Cache hits included): 2 x FETCH_SIZE + WRITE_SIZE",
                                    "traffic_fetch_size_raw": t["fetch_raw"], "traffic_write_size": t["write"],
                                    "traffic_over_algorithmic": t["bytes"] / (bytes_per_cw * B)})
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_encode(cfg, args.cpu_seconds)
    emit(res)


def pmc_traffic(config, log2b, kernel_substr):
    """L2<->fabric bytes per launch of the dominant kernel from the PMC counters FETCH_SIZE / WRITE_SIZE, each collected in
    its own rocprofv3 pass (no tracing) around a child run of this script, as MI355X_MICROARCH.md prescribes:
    the counters are in KiB, count Infinity-Cache hits, and on gfx950 FETCH_SIZE tallies every 128-B request as 64 B
    (reported value x 2 = bytes).  Returns None if rocprofv3 is unavailable."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_CHILD="1")
            subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", sys.executable,
                            os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "2", "--warmup", "0",
                            "--no-cpu-baseline", "--no-pmc-traffic", "--log2-batch", str(log2b)],
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=300, check=True)
            vals = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            out[counter] = float(np.mean(vals)) * 1024.0
        except Exception as e:  # noqa: BLE001
            print(f"[bench] PMC pass {counter} failed: {e}", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return {"fetch_raw": out["FETCH_SIZE"], "write": out["WRITE_SIZE"],
            "bytes": 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]}


def pmc_issue(config, log2b, kernel_substr, codewords):
    """Instruction mix and wait share of the dominant kernel from one rocprofv3 --pmc pass over the SQ counters (its own
    child run, no tracing).  Per codeword: wave-instructions issued; salu_frac / valu_frac: share of the scalar-unit
    (SALU + branch) / vector-ALU instructions among all issued instructions; wait_frac: SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
             "SQ_ACTIVE_INST_ANY"]
    tmp = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_CHILD="1")
        subprocess.run(["rocprofv3", "--pmc"] + names + ["--output-format", "csv", "-d", tmp, "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-pmc-traffic", "--log2-batch", str(log2b)],
                       cwd="/tmp", env=env, capture_output=True, text=True, timeout=300, check=True)
        agg, n = {}, {}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_substr in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
        if "SQ_INSTS_VALU" not in agg:
            return None
        per = {k: agg[k] / n[k] / codewords for k in agg}  # per launch, per codeword
        insts = sum(per.get(k, 0.0) for k in names[:5])
        return {"valu_per_codeword": per["SQ_INSTS_VALU"], "salu_per_codeword": per["SQ_INSTS_SALU"],
                "branch_per_codeword": per["SQ_INSTS_BRANCH"], "lds_per_codeword": per["SQ_INSTS_LDS"],
                "vmem_per_codeword": per["SQ_INSTS_VMEM"],
                "salu_frac": (per["SQ_INSTS_SALU"] + per["SQ_INSTS_BRANCH"]) / insts, "valu_frac": per["SQ_INSTS_VALU"] / insts,
                "wait_frac": per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"],
                "active_frac": per["SQ_ACTIVE_INST_ANY"] / per["SQ_WAVE_CYCLES"],
                # how far from the one-wave issue wall the kernel runs: a wave alone on its SIMD issues one instruction of any
                # kind per 4 cycles (MI355X_MICROARCH.md), SQ_WAVE_CYCLES counts quad-cycles, so instructions / SQ_WAVE_CYCLES
                # = (instructions x 4 cycles) / wave cycles.  Meaningful for the one-wave-per-SIMD kernels (lane / pair).
                "issue_bound_frac": insts / per["SQ_WAVE_CYCLES"],
                "source": "rocprofv3 --pmc " + " ".join(names) + " (one child pass, 1 step)"}
    except Exception as e:  # noqa: BLE001
        print(f"[bench] PMC pass (SQ) failed: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(cfg, seconds):
    """the reference's own CPU decoder (oracle/_ref, unmodified library) if it travelled with the repo,
    else the oracle port -- on a bounded sample of the same workload; reported, never the target."""
    cores = len(os.sched_getaffinity(0))
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    rows = 4096
    if os.path.exists(harness):
        try:
            out = subprocess.run([harness, "time", str(cfg["N"]), str(cfg["K"]), str(cfg["L"]), cfg.get("ctor", "BB"), str(cfg["dsnr"]),
                                  str(cfg["crc"]), cfg["crc_type"], str(cfg["ebn0"]), str(rows), str(cores), str(seconds)]
                                 + (["char"] if cfg.get("dectype") == "char" else []),
                                 capture_output=True, text=True, timeout=seconds * 6 + 120, check=True)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            return {"value": r["cw_per_s"], "unit": "codewords/s", "cores": cores, "kind": "reference",
                    "sample": f"{rows} distinct AWGN codewords cycled for {r['seconds']:.1f} s, one reference "
                              f"decoder object per thread ({cores} threads), {r['codewords']} decodes"}
        except Exception as e:  # noqa: BLE001
            print(f"[bench] reference harness unavailable ({e}); timing the oracle port", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    frozen = config_frozen_set(cfg)
    rng = np.random.default_rng(0)
    llr = rng.normal(1.0, 0.8, (256, cfg["N"])).astype(np.float32)
    if cfg.get("dectype") == "char":
        llr = np.clip(np.rint(llr * 8.0), -127, 127).astype(np.float32)
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        ob.decode_batch(cfg["N"], cfg["L"], frozen, cfg["crc"], llr, True, cfg.get("dectype", "float"))
        n += llr.shape[0]
    el = time.time() - t0
    return {"value": n / el, "unit": "codewords/s", "cores": 1, "kind": "port",
            "sample": f"{n} decodes of 256 noisy rows by the scalar C oracle in {el:.1f} s"}


def main():
    # stdout carries exactly ONE line, the JSON result: everything libraries print while the benchmark runs (RCCL
    # writes a version banner to stdout when the first communicator is created) goes to stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    args = parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        os.dup2(real_stdout, 1)          # the ranks inherit the real stdout: rank 0 writes the JSON line to it
        sys.exit(self_launch(args))
    if args.launch_check:
        info = {"launch_check": True, "rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")),
                "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "master": os.environ.get("MASTER_ADDR", ""),
                "gpus": args.gpus}
        if os.environ.get("BENCH_LAUNCH_CHECK_FAIL_RANK") == str(info["rank"]):
            sys.exit(3)
        if os.environ.get("BENCH_LAUNCH_CHECK_HANG_RANK") == str(info["rank"]):
            time.sleep(3600)             # (test of --launch-timeout: a rank that never comes back)
        if info["rank"] == 0:
            emit(info)
        return
    if args.kernel_name:  # (host side only: which kernel the library picks for this config -- tools/collect_profiles.sh)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "polar-codes_amd"))
        import pypolar_amd as _pp
        _cfg = CONFIGS[args.config]
        if _cfg["L"] == 0:
            os.write(real_stdout, b"\n")
            return
        from pypolar_amd.construction import frozen_bits as _fb
        _d = _pp.PolarDecoder(_cfg["N"], _cfg["L"], _fb(_cfg["N"], _cfg["K"], _cfg["dsnr"], _cfg.get("ctor", "BB")), "char" if _cfg.get("dectype") == "char" else "float")
        os.write(real_stdout, (_d.kernel_name(_cfg["L"] > 1) + "\n").encode())
        return
    heavy_imports()
    if args.pmc_traffic is None:  # default: on for the single-GPU run when the profiler is there (and not inside its own child pass)
        import shutil
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        args.pmc_traffic = (args.gpus == 1 and shutil.which("rocprofv3") is not None and not profiled
                            and not os.environ.get("BENCH_PMC_CHILD"))

    cfg = dict(CONFIGS[args.config])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # BENCH_BACKEND=gloo: rehearsal of the N > 1 plumbing on a box with fewer GPUs than ranks (ranks share
    # devices, the gather goes through host memory); the driver's runs use the default, nccl (= RCCL).
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # BENCH_FORCE_GATHER=1: run the N > 1 exchange code with a single rank (rehearsal of the stream overlap on one GPU)
    exchange = world > 1 or bool(os.environ.get("BENCH_FORCE_GATHER"))
    if exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.config.startswith("enc"):
        def enc_barrier():
            if world > 1:
                torch.distributed.barrier(device_ids=[dev_index]) if backend == "nccl" else torch.distributed.barrier()
            torch.cuda.synchronize()
        bench_encode(args, cfg, dev, dev_index, rank, world, enc_barrier, emit)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.config.startswith("dep"):
        def dep_barrier():
            if world > 1:
                torch.distributed.barrier(device_ids=[dev_index]) if backend == "nccl" else torch.distributed.barrier()
            torch.cuda.synchronize()
        bench_depuncture(args, cfg, dev, dev_index, rank, world, dep_barrier, emit)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    log2b = args.log2_batch if args.log2_batch is not None else cfg["log2_batch"]
    B = 1 << log2b                                   # per GPU (weak scaling: fixed work per GPU)
    N, K, L = cfg["N"], cfg["K"], cfg["L"]
    nb = (K + 7) // 8
    frozen = config_frozen_set(cfg)
    chr_dec = cfg.get("dectype") == "char"
    dec = pypolar_amd.PolarDecoder(N, L, frozen, "char" if chr_dec else "float", device=dev_index)
    dec.setErrorDetection(cfg["crc"], cfg["crc_type"])
    # the kernel this decoder launches, as the library reports it (pcd_kernel_name): the PMC passes filter their rows by it
    kernel_name = dec.kernel_name(L > 1)

    llr, info = make_batch(cfg, frozen, B, dev, seed=1234 + rank)
    if chr_dec:  # int8 rows resident in HBM: round(8 y), clipped to +-127
        llr = torch.clamp(torch.round(llr * 8.0), -127, 127).to(torch.int8)
    out = torch.empty((B, nb), dtype=torch.uint8, device=dev)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    # N > 1: the only exchange of the path -- decoded bytes + verdict of every rank (RCCL all-gather over xGMI).
    # It runs on its own stream, double buffered, so that the gather of step k overlaps the decode of step k + 1;
    # the closing barrier synchronises the whole device, so every gather lies inside the timed region.
    packed = [torch.empty((B, nb + 1), dtype=torch.uint8, device=dev) for _ in range(2)] if exchange else None
    gathered = [torch.empty((B * world, nb + 1), dtype=torch.uint8, device=dev) for _ in range(2)] if exchange else None
    comm = torch.cuda.Stream(device=dev) if exchange and backend == "nccl" else None
    ev_ready = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    state = {"k": 0}

    def step():
        dec.decode_batch(llr, out=out, ok=ok)
        if not exchange:
            return None
        b = state["k"] & 1
        state["k"] += 1
        main = torch.cuda.current_stream(dev)
        if state["k"] > 2:
            main.wait_event(ev_done[b])          # the gather that last read packed[b] has finished
        packed[b][:, :nb].copy_(out)
        packed[b][:, nb].copy_(ok)
        if backend != "nccl":  # rehearsal path: gloo gathers host tensors
            return sharding.gather_decoded(packed[b].cpu(), equal_rows=True)
        ev_ready[b].record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(ev_ready[b])
            res = sharding.gather_decoded(packed[b], equal_rows=True, out=gathered[b])  # one collective, no host sync
            ev_done[b].record(comm)
        return res

    def barrier():
        if exchange:
            if backend == "nccl":
                torch.distributed.barrier(device_ids=[dev_index])
            else:
                torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dec.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if exchange:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    kms = dec.kernel_times_ms()
    dec.set_timing(False)

    # sanity outside the timed region: decoded payloads vs what was sent
    okf = float(ok.float().mean().item())
    match = float((out == info).all(dim=1).float().mean().item())

    if rank == 0:
        total_cw = B * world * args.steps
        cw_s = total_cw / elapsed
        payload_bits = K - cfg["crc"] if cfg["crc"] != 11 else K - 16
        k_ms = float(np.mean(kms)) if len(kms) else float("nan")
        bytes_per_cw = (N if chr_dec else 4 * N) + nb   # fp32 (int8: SURVEY 8f-3) row in + packed bytes out
        achieved = bytes_per_cw * B / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "codewords/sec + info-Gbit/s, N=1024 K=512 SCL L=8, batch=2^20" if args.config == "cfg3"
                      else f"codewords/sec + info-Gbit/s, {cfg['name']}",
            "value": cw_s, "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i8" if chr_dec else "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + (" per GPU" if world > 1 else ""), "block_length": N,
                       "info_bits": K, "list_size": L, "detector": f"{cfg['crc_type']}-{cfg['crc']}",
                       "ebn0_db": cfg["ebn0"], "codewords_per_gpu": B,
                       "frozen_set": "5G reliability table" if cfg.get("ctor") == "5G" else f"BB dSNR {cfg['dsnr']} dB",
                       "parallelism": f"rows sharded over {world} GPU(s), all-gather of decoded bytes" if world > 1
                                      else "single GPU"},
            "info_gbit_per_s": cw_s * K / 1e9, "payload_gbit_per_s": cw_s * payload_bits / 1e9,
            "detector_pass_fraction": okf, "payload_match_fraction": match,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel_name,
                         "kernel_ms_avg": k_ms, "algorithmic_bytes_per_codeword": bytes_per_cw,
                         "codewords_per_launch": B},
        }
        if world == 1 and args.pmc_traffic:
            t = pmc_traffic(args.config, log2b, res["roofline"]["kernel"])
            if t:
                res["roofline"].update({"traffic": t["bytes"], "traffic_unit": "bytes per launch at the L2<->fabric "
                                        "interface (Infinity-Cache hits included): 2 x FETCH_SIZE + WRITE_SIZE",
                                        "traffic_fetch_size_raw": t["fetch_raw"], "traffic_write_size": t["write"],
                                        "traffic_over_algorithmic": t["bytes"] / (bytes_per_cw * B)})
            iss = pmc_issue(args.config, log2b, res["roofline"]["kernel"], B)
            if iss:
                res["roofline"]["issue"] = iss
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        emit(res)
    if exchange:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
else:
    heavy_imports()  # imported as a library (tools/*.py use make_batch / CONFIGS): no launcher logic, load everything
