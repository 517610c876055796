#!/usr/bin/env python3
"""bench.py -- the AGMV hot path on MI355X: frames/s encode+decode, and the HBM roofline of the
dominant kernel, next to the reference CPU path timed on the same box.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5] [--frames T] [--backend nccl|gloo]
                  [--no-cpu-baseline] [--no-normal-heavy] [--no-secondary]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one batch of synthetic input already resident in HBM:
  encode  k_encode            loops A+B of AGMV_EncodeFrame for every frame of the clip
  decode  parser + k_decode   AGMV_DecodeFrameChunk's parse + reconstruct for every frame
(the LZSS stage, file I/O and the palette build are host work by design and are not in the step).

Workload (BASELINE.json configs):
  c3 (default)  1024-frame 1920x1080 agmv_synth_v1 clip per GPU, AGMV_HIGH_QUALITY palette, OPT_III
                (512 colours) -- the configuration the roofline target is quoted on
  c2            212 source frames 320x240 -> the 156 encoded frames of AGMV_EncodeAGMV/OPT_III (light PDIFS)
  c5            1280x720 clip (use --frames; 8192 frames need ~70 GB of HBM per GPU)
Multi-GPU: weak scaling by default -- every rank encodes+decodes its own T-frame clip (frames r*T .. r*T+T-1 of one
long clip), value = N*T*K / max-over-ranks time.  --scaling strong: ONE T-frame clip (configs 4 and 5) is split over the
ranks by GOP range (libagmv_amd.shard.gop_ranges), value = T*K / max-over-ranks time.  Either way the only collectives are
outside the timed region: the palette histogram all-reduce before, and the final gather of the per-frame bitstreams to
rank 0 after (timed separately: final_gather_ms).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c5"])
    ap.add_argument("--no-normal-heavy", action="store_true",
                    help="skip the k_encode run on a NORMAL-heavy variant of the clip (roofline_normal_heavy; it runs AFTER the timed region -- "
                         "use this switch for a profile that should hold the headline launches only)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the 320x240 lines (secondary.c2: the 156 encoded frames of config 2; secondary.c2_large: 8192 frames), "
                         "which also run after the timed region")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo lets several ranks share ONE card to rehearse the multi-rank path)")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: the workload's)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own clip of --frames frames; strong: ONE clip of --frames frames split over the ranks by GOP range")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames of the CPU baseline sample")
    return ap.parse_args()


def workload(args):
    if args.workload == "c3":
        return dict(name="c3: %d-frame 1920x1080 agmv_synth_v1, HIGH_QUALITY palette, OPT_III (512 colours)" % (args.frames or 1024),
                    W=1920, H=1080, T=args.frames or 1024, quality=1, pdifs=False)
    if args.workload == "c5":
        return dict(name="c5: %d-frame 1280x720 agmv_synth_v1 stream, HIGH_QUALITY palette, OPT_III" % (args.frames or 1024),
                    W=1280, H=720, T=args.frames or 1024, quality=1, pdifs=False)
    return dict(name="c2: 212-frame 320x240 agmv_synth_v1 -> 156 encoded frames (AGMV_EncodeAGMV light PDIFS), "
                     "LOW_QUALITY palette, OPT_III", W=320, H=240, T=args.frames or 212, quality=3, pdifs=True)


def host_lib():
    from libagmv_amd import build
    build.build()
    L = C.CDLL(os.path.join(ROOT, "libagmv_amd", "libagmv.so"))
    L.AGMV_BuildPalette.restype = None
    L.AGMV_BuildPalette.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return L


def cpu_baseline(wl, p0, p1, frames_np, gpu_bits, gpu_pix=None, gpu_decode=None):
    """the reference's own compiled code (oracle/_ref, kind 'reference') or, where that build is absent, the oracle
    restatement (kind 'port') on a bounded sample of the same clip, one host thread.
      encode leg: AGMV_FindNearestEntry per pixel + AGMV_Assemble{I,P}FrameBitstream (loops A+B of AGMV_EncodeFrame);
      decode leg: the reference's AGMV_DecodeFrameChunk (src/agmv_decode.c:145-410) on a .agmv file holding the sample's
                  chunks -- the GPU's bitstreams, which the encode leg has just shown equal to the reference's, compressed
                  by libagmv.so's exact LZSS (the reference's own O(n * 65535) LZSS would take minutes for the sample).
                  Its time includes the reference's LZSS-decode stage, which stays on the host on our side too.
    `legs` says which code ran each leg.  Also a parity check: the bitstreams must equal the GPU's, and the pixels must equal
    what the GPU (gpu_decode) makes of the SAME decompressed bytes -- the reference's LZSS stores floor(bits / 8) as csize, so
    a decompressed stream can come out a byte short and the frame then keeps some of the previous frame's pixels
    (src/agmv_decode.c:229-232): part of the format, and exactly what agmv_hip_decode_bitstreams_dev reproduces."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracles as O
    W, H = wl["W"], wl["H"]
    kind = "reference" if O.have_ref() else "port"
    enc = (O.RefEncoder if kind == "reference" else O.OracleEncoder)(W, H, True, p0, p1)
    t_enc = t_dec = 0.0
    bits = []
    for k, f in enumerate(frames_np):
        t0 = time.perf_counter()
        b = enc.encode(f)
        t_enc += time.perf_counter() - t0
        bits.append(b)
        if gpu_bits is not None and not (len(b) == len(gpu_bits[k]) and (b == gpu_bits[k]).all()):
            raise SystemExit("bench: GPU bitstream of frame %d differs from the %s CPU path" % (k, kind))
    n = len(frames_np)
    dec_kind = kind
    if kind == "reference":
        import tempfile
        import hostlib as HL
        L = O.ref()
        a = L.refshim_create(W, H, O.OPT_III, O.LZSS, np.ascontiguousarray(p0, np.uint32), np.ascontiguousarray(p1, np.uint32))
        tmp = os.path.join(tempfile.mkdtemp(prefix="agmv_bench_"), "sample.agmv")
        L.refshim_write_header(a, tmp.encode())
        L.refshim_destroy(a)
        with open(tmp, "ab") as fh:
            for k, b in enumerate(bits):
                comp, cs = HL.lzss(b)
                fh.write(b"AGFC" + np.array([k + 1, len(b), cs], "<u4").tobytes() + comp.tobytes() + b"\xff" * 8)
        err = C.c_int(0)
        w_, h_, n_, v_ = (C.c_uint32(0) for _ in range(4))
        d = L.refshim_decoder_open(tmp.encode(), C.byref(err), C.byref(w_), C.byref(h_), C.byref(n_), C.byref(v_))
        if err.value != 0:
            raise SystemExit("bench: the reference decoder rejected the sample file (error %d)" % err.value)
        pix = np.zeros(W * H, np.uint32)
        ref_pix, ref_bits, ref_bpos = [], [], []
        bp = C.c_uint32(0)
        for k in range(n):
            t0 = time.perf_counter()
            e = L.refshim_decoder_next(d, pix.ctypes.data_as(C.c_void_p), None, None, C.byref(bp))
            t_dec += time.perf_counter() - t0
            if e != 0:
                raise SystemExit("bench: the reference decoder failed on frame %d (error %d)" % (k, e))
            bs = np.zeros(bp.value + 16, np.uint8)
            L.refshim_decoder_bitstream(d, bs, len(bs))           # what its LZ stage left in the persistent buffer (+ the stale bytes behind)
            ref_pix.append(pix.copy()); ref_bits.append(bs); ref_bpos.append(bp.value)
        if gpu_decode is not None:
            got = gpu_decode(ref_bits, ref_bpos)
            for k in range(n):
                if not (got[k] == ref_pix[k]).all():
                    raise SystemExit("bench: GPU decode of frame %d differs from the reference's AGMV_DecodeFrameChunk" % k)
        L.refshim_decoder_close(d)
        os.unlink(tmp)
    else:
        dec = O.OracleDecoder(W, H, True, p0, p1)
        for k, b in enumerate(bits):
            t0 = time.perf_counter()
            pix = dec.decode(b)
            t_dec += time.perf_counter() - t0
            if gpu_pix is not None and not (pix == gpu_pix[k]).all():
                raise SystemExit("bench: GPU decode of frame %d differs from the CPU decoder" % k)
    return {"value": round(n / (t_enc + t_dec), 4), "unit": "frames/s", "cores": 1, "kind": kind,
            "legs": {"encode": kind, "decode": dec_kind},
            "sample": "first %d encoded frames of the same clip (%dx%d): AGMV_FindNearestEntry per pixel + "
                      "Assemble{I,P}FrameBitstream (%.2f s/frame), then AGMV_DecodeFrameChunk = LZSS-decode + parse + reconstruct "
                      "(%.4f s/frame); the LZSS ENCODER excluded on both sides" % (n, W, H, t_enc / n, t_dec / n),
            "host_cores_available": os.cpu_count()}


def main():
    args = parse_args()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())        # rehearsal: the ranks share the cards there are
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        raise SystemExit("bench: for --gpus N > 1 launch with torch.distributed.run (see the docstring)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from libagmv_amd import AgmvHip
    hip = AgmvHip(local_rank)
    wl = workload(args)
    W, H, T = wl["W"], wl["H"], wl["T"]
    npx = W * H

    # ------------------------------------------------------------------ setup (untimed)
    first_fc = 0
    if wl["pdifs"]:
        if args.scaling == "strong":
            raise SystemExit("bench: --scaling strong is for the resident workloads (c3, c5)")
        # AGMV_EncodeAGMV light schedule (reference src/agmv_encode.c:2727-2752): per 4 inputs i..i+3 encode
        # f(i), midpoint(f(i+1), f(i+2)), f(i+3); stop when i+4 >= end.  212 inputs -> 156 encoded frames.
        src = hip.synth_dev(W, H, rank * T + 1, T, device=dev)          # frames numbered 1..T like the BMP files
        pick = []
        i = 1
        while i <= T:
            pick += [(i, -1), (i + 1, i + 2), (i + 3, -1)]
            i += 4
            if i + 4 >= T:
                break
        frames = torch.empty((len(pick), H, W), dtype=torch.int32, device=dev)
        for k, (a, b) in enumerate(pick):
            frames[k] = src[a - 1] if b < 0 else hip.interp_dev(src[a - 1], src[b - 1])
        hist_src = src
    elif args.scaling == "strong":
        from libagmv_amd import shard
        lo, hi = shard.gop_ranges(T, world)[rank]              # this rank's GOPs of the ONE clip
        if hi <= lo:
            raise SystemExit("bench: --scaling strong needs at least one GOP per rank (%d frames, %d ranks)" % (T, world))
        first_fc = lo
        frames = hip.synth_dev(W, H, lo, hi - lo, device=dev)
        hist_src = frames
    else:
        frames = hip.synth_dev(W, H, rank * T, T, device=dev)
        hist_src = frames
    n_enc = frames.shape[0]
    hist = hip.histogram_dev(hist_src.reshape(-1), wl["quality"])
    if dist is not None:
        dist.all_reduce(hist)                                   # one palette for the whole job (pass 1 over all frames)
    torch.cuda.synchronize()
    hist_np = hist.cpu().numpy().view(np.uint32)
    p0 = np.zeros(256, np.uint64)
    p1 = np.zeros(256, np.uint64)
    host_lib().AGMV_BuildPalette(hist_np.ctypes.data, wl["quality"], 3, p0.ctypes.data, p1.ctypes.data)   # OPT_III
    p0 = p0.astype(np.uint32)
    p1 = p1.astype(np.uint32)
    hip.set_palette(p0, p1, True)
    hip.enable_timing(True)

    stride = hip.max_usize(W, H)
    out = torch.empty((n_enc, stride), dtype=torch.uint8, device=dev)
    sizes = torch.empty(n_enc, dtype=torch.int32, device=dev)
    nblk = npx // 16
    nent = torch.empty(n_enc, dtype=torch.int32, device=dev)
    dec = torch.empty((n_enc, H, W), dtype=torch.int32, device=dev)

    def step():
        hip.encode_dev(frames, n_enc, W, H, first_fc, out=out, sizes=sizes)
        hip.decode_bitstreams_dev(out, sizes, n_enc, W, H, first_fc, out=dec, nentered=nent)   # parser + k_decode, entry bitmaps in between

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    hip.check()

    # ------------------------------------------------------------------ timed region: exactly K steps
    k_ms = {"encode": [], "parse": [], "decode": []}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # per-kernel durations from the HIP events the library recorded on this stream (reading them waits for the
        # step's last kernel, which the next step depends on anyway)
        for i, k in enumerate(("encode", "parse", "decode")):
            k_ms[k].append(hip.last_kernel_ms(i))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    hip.check()

    # ------------------------------------------------------------------ results
    usz = sizes.cpu().numpy().astype(np.int64)
    assert (nent.cpu().numpy() == nblk).all(), "decoder did not reach every block of a clean stream"
    alg_bytes = 4 * npx * n_enc + int(usz.sum())               # 4*W*H + usize per frame (same for decode)
    enc_ms = float(np.mean(k_ms["encode"]))
    par_ms = float(np.mean(k_ms["parse"]))
    dec_ms = float(np.mean(k_ms["decode"]))
    achieved = alg_bytes / (enc_ms * 1e-3) / 1e9

    # final gather of the job's bitstreams to rank 0 (the only data collective; outside the timed region)
    gather_ms, gather_err, gather_hung = None, None, False
    if dist is not None:
        torch.cuda.synchronize()
        from libagmv_amd import shard
        import threading
        box, done = {}, threading.Event()

        def final_gather():
            # outside the timed region, and on a watchdog: the RCCL send/recv path has only ever been rehearsed with gloo (no
            # multi-GPU box in the build rounds) -- a failure or a hang here must not cost the job its line
            try:
                torch.cuda.set_device(local_rank)
                g0 = time.perf_counter()
                gathered = shard.gather_bitstreams(dist, sizes, shard.pack_frames(out, sizes, hip=hip), dst=0)
                if rank == 0:
                    assert sum(int(s.numel()) for s, _ in gathered) == (T if args.scaling == "strong" else world * n_enc)
                torch.cuda.synchronize()
                box["ms"] = (time.perf_counter() - g0) * 1e3
            except Exception as e:                             # noqa: BLE001 -- reported in the line, not hidden
                box["err"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            finally:
                done.set()

        threading.Thread(target=final_gather, daemon=True).start()
        if not done.wait(float(os.environ.get("AGMV_BENCH_GATHER_TIMEOUT", "120"))):
            gather_err, gather_hung = "no completion within the watchdog's time", True
        else:
            gather_ms, gather_err = box.get("ms"), box.get("err")

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%dx%dx%d" % (W, H, n_enc)
                if key in tj.get("k_encode", {}):
                    traffic = tj["k_encode"][key]
            except Exception:
                traffic = None
        res = {
            "metric": "frames/s encode+decode (AGMV hot path: quantise+classify+assemble, parse+reconstruct), synthetic",
            "value": round((T if args.scaling == "strong" else world * n_enc) * args.steps / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl["name"], "frames_per_gpu": int(n_enc), "width": W, "height": H,
                       "palette": "512 colours, reference histogram build (GPU histogram + host pick)",
                       "parallelism": ("one %d-frame clip split by GOP range over %d rank(s)" % (T, world) if args.scaling == "strong" else
                                       "%d frames per rank, %d rank(s)" % (int(n_enc), world)) + ", no data-path collective",
                       "mean_usize_bytes": float(usz.mean())},
            "roofline": {"bound": "hbm", "kernel": "k_encode", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(enc_ms, 4)},
            "kernels_ms": {"k_encode": round(enc_ms, 4), "k_parse_*": round(par_ms, 4), "k_decode+k_fixup": round(dec_ms, 4)},
            "decode_roofline": {"achieved": round(alg_bytes / (dec_ms * 1e-3) / 1e9, 1),
                                "frac": round(alg_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "with_parser_frac": round(alg_bytes / ((dec_ms + par_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if gather_ms is not None:
            res["final_gather_ms"] = round(gather_ms, 3)
        if gather_err is not None:
            res["final_gather_error"] = gather_err
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = args.cpu_frames or (8 if npx > 500000 else 32)
            n_cpu = min(n_cpu, n_enc)
            f_np = frames[:n_cpu].cpu().numpy().view(np.uint32)
            gpu_bits = [out[f, :int(usz[f])].cpu().numpy() for f in range(n_cpu)]
            gpu_pix = [dec[f].cpu().numpy().view(np.uint32).reshape(-1) for f in range(n_cpu)]
            def gpu_decode(bits_list, bpos_list):              # the same decompressed bytes through the GPU path
                stride_s = (max(len(b) for b in bits_list) + 255) & ~255
                slab = np.zeros((len(bits_list), stride_s), np.uint8)
                for k, b in enumerate(bits_list):
                    slab[k, :len(b)] = b
                px = hip.decode_bitstreams_dev(torch.from_numpy(slab).to(dev), torch.tensor(bpos_list, dtype=torch.int32, device=dev),
                                               len(bits_list), W, H, first_fc)
                torch.cuda.synchronize()
                return px.cpu().numpy().view(np.uint32).reshape(len(bits_list), -1)
            res["cpu_baseline"] = cpu_baseline(wl, p0, p1, list(f_np), gpu_bits, gpu_pix, gpu_decode)
        if not args.no_normal_heavy and world == 1 and npx > 500000 and args.scaling == "weak":     # (after the timed region: it re-uses the bitstream slab)
            res["roofline_normal_heavy"] = normal_heavy_leg(torch, hip, frames, W, H, first_fc, out, sizes)
        if not args.no_secondary and world == 1 and args.workload == "c3" and args.scaling == "weak":
            del dec
            res["secondary"] = secondary_legs(torch, local_rank, dev)
        print(json.dumps(res))
    if dist is not None:
        if gather_hung or gather_err is not None:              # the process group is in an unknown state: leave without its teardown
            sys.stdout.flush()
            os._exit(0)
        dist.barrier()
        dist.destroy_process_group()
    hip.close()


def normal_heavy_leg(torch, hip, frames, W, H, first_fc, out, sizes, n=256):
    """k_encode on content that defeats FILL / COPY: the first n frames of the clip with 3 bits of noise on every channel
    (NORMAL blocks almost everywhere, ~3x the bitstream, look-ups that rarely share a table line).  Reported beside the
    headline roofline line, never instead of it; outside the timed region."""
    n = min(n, frames.shape[0])
    g = torch.Generator(device=frames.device)
    g.manual_seed(7)
    clip = frames[:n].clone()
    for sh in (0, 8, 16):
        clip ^= torch.randint(0, 8, clip.shape, dtype=torch.int32, device=frames.device, generator=g) << sh
    ms = []
    for i in range(5):
        hip.encode_dev(clip, n, W, H, first_fc, out=out[:n], sizes=sizes[:n])
        if i >= 2:
            ms.append(hip.last_kernel_ms(0))
    hip.check()
    alg = 4 * W * H * n + int(sizes[:n].cpu().numpy().astype(np.int64).sum())
    t = float(np.mean(ms))
    return {"kernel": "k_encode", "achieved": round(alg / (t * 1e-3) / 1e9, 1), "frac": round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "avg_launch_ms": round(t, 4), "algorithmic_bytes_per_launch": alg,
            "sample": "first %d frames of the clip XOR 3 bits of noise per channel (mean usize %d bytes)" % (n, int(sizes[:n].float().mean()))}


def secondary_legs(torch, local_rank, dev):
    """north_star's other resolution, 320x240, after the timed region: the clip of config 2 (212 source frames -> the 156
    encoded frames of AGMV_EncodeAGMV's light PDIFS schedule) and a machine-filling batch (8192 frames).  Kernel times from
    HIP events (median of 5 launches after 2), fractions of the 8 TB/s HBM peak on algorithmic bytes like the headline."""
    from libagmv_amd import AgmvHip
    W, H = 320, 240
    npx = W * H
    h2 = AgmvHip(local_rank)
    out = {}
    try:
        src = h2.synth_dev(W, H, 1, 212, device=dev)
        pick, i = [], 1
        while i <= 212:
            pick += [(i, -1), (i + 1, i + 2), (i + 3, -1)]
            i += 4
            if i + 4 >= 212:
                break
        c2 = torch.empty((len(pick), H, W), dtype=torch.int32, device=dev)
        for k, (a, b) in enumerate(pick):
            c2[k] = src[a - 1] if b < 0 else h2.interp_dev(src[a - 1], src[b - 1])
        hist = h2.histogram_dev(src.reshape(-1), 3)
        torch.cuda.synchronize()
        p0 = np.zeros(256, np.uint64)
        p1 = np.zeros(256, np.uint64)
        hist_np = hist.cpu().numpy().view(np.uint32)            # (kept alive across the call: ctypes takes a bare pointer)
        host_lib().AGMV_BuildPalette(hist_np.ctypes.data, 3, 3, p0.ctypes.data, p1.ctypes.data)
        h2.set_palette(p0.astype(np.uint32), p1.astype(np.uint32), True)
        h2.enable_timing(True)
        big = h2.synth_dev(W, H, 0, 8192, device=dev)
        for name, clip in (("c2", c2), ("c2_large", big)):
            n = clip.shape[0]
            bits = torch.empty((n, h2.max_usize(W, H)), dtype=torch.uint8, device=dev)
            sizes = torch.empty(n, dtype=torch.int32, device=dev)
            nent = torch.empty(n, dtype=torch.int32, device=dev)
            pix = torch.empty((n, H, W), dtype=torch.int32, device=dev)
            ms = {"e": [], "p": [], "d": []}
            wall = []
            for it in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                h2.encode_dev(clip, n, W, H, 0, out=bits, sizes=sizes)
                h2.decode_bitstreams_dev(bits, sizes, n, W, H, 0, out=pix, nentered=nent)
                torch.cuda.synchronize()
                if it >= 2:
                    wall.append(time.perf_counter() - t0)
                    ms["e"].append(h2.last_kernel_ms(0)); ms["p"].append(h2.last_kernel_ms(1)); ms["d"].append(h2.last_kernel_ms(2))
            h2.check()
            assert (nent.cpu().numpy() == npx // 16).all()
            alg = 4 * npx * n + int(sizes.cpu().numpy().astype(np.int64).sum())
            e, p, d = (float(np.median(ms[k])) for k in ("e", "p", "d"))
            fr = lambda t: round(alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            out[name] = {"frames": int(n), "width": W, "height": H, "algorithmic_bytes": alg,
                         "kernels_ms": {"k_encode": round(e, 4), "parser": round(p, 4), "k_decode+k_fixup": round(d, 4)},
                         "encode_frac": fr(e), "decode_frac": fr(d), "parse_decode_frac": fr(p + d),
                         "frames_per_s_encode_decode": round(n / float(np.median(wall)), 1)}
            del bits, pix
    finally:
        h2.close()
    return out


if __name__ == "__main__":
    main()
