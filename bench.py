#!/usr/bin/env python3
"""Benchmark of the hot path qvrcnn::forward_blu (inference/qvrcnn.cu:168-242) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of forward_blu over one batch of synthetic Y frames that are already resident
in HBM (default: 512 frames of 1920x1080 per GPU, QP37 model: BASELINE.json configs[1]; the fused
kernels walk them 16 frames per launch).  The batch is that large so that K = 20 steps keep the
device busy for about a second: shorter bursts ride the boost clock and overstate the sustained
rate.  Before the W warm-up steps the loop runs untimed until the device has been busy for
--prewarm seconds (clock ramp).

With N > 1 frames are sharded across one process per GPU with no data-path collective (weak
scaling); the only communication is the barrier / max-reduction that brackets the timed region.
Started plainly (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script launches
its N ranks itself through torch.distributed.run, before anything touches the GPU; started by
torch.distributed.run it is one rank.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import platform
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 5000.0   # dense int8 MFMA peak of MI355X: 2x the ~2.5 PFLOP/s bf16 dense figure
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--qp", type=int, default=37)
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (0: about 1 GiB of frames, 512 at 1080p)")
    ap.add_argument("--group", type=int, default=0, help="frames per kernel launch (0 = engine default)")
    ap.add_argument("--prewarm", type=float, default=2.0, help="seconds of untimed launches before the warm-up steps")
    ap.add_argument("--exact", action="store_true",
                    help="force the float-pipeline kernels (what a model with out-of-range parameters runs on)")
    ap.add_argument("--wide", action="store_true",
                    help="force kernel A's clamped integer form (what a model whose C2 accumulator bounds pass 2^24 runs on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-shared-gpus", action="store_true",
                    help="let ranks share a device when fewer than --gpus are present (rehearsal only)")
    ap.add_argument("--e2e", action="store_true", help="(default at N=1; kept for old command lines)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the PCIe-inclusive legs (the timed region of inference/kernel.cu:89-101: 20 blocking frames, "
                         "then one 256-frame pinned clip through the pipelined entry point)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run job.
    Nothing in this process has touched the GPU yet, and it only waits for the child."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(qp, h, w):
    """float32 restatement of training/model.py (model_blu) on the host cores, bounded sample."""
    import torch
    from oracle import float_model
    from qcnn_gpu_amd import quant_tables, synth
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    m = synth.synth_model(qp)
    _, anc = synth.synth_frames(h, w, qp)
    ws, bs, blu = float_model.dequantise(m, quant_tables.QUANT_ROWS[qp])
    torch.set_num_threads(cores)
    float_model.forward_float(ws, bs, blu, anc[0][:64, :64].copy())     # warm-up (configs[0] shape)
    t0 = time.perf_counter()
    float_model.forward_float(ws, bs, blu, anc[0])
    first = time.perf_counter() - t0
    reps = max(1, min(30, int(10.0 / max(first, 1e-3))))   # about 10 s of CPU work on all cores
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        float_model.forward_float(ws, bs, blu, anc[0])
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    # one thread: a quarter-height band of the same frame, once (a few seconds)
    band = np.ascontiguousarray(anc[0][: max(h // 4, min(h, 64))])
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    float_model.forward_float(ws, bs, blu, band)
    one = time.perf_counter() - t0
    torch.set_num_threads(cores)
    return {"value": round(h * w / med / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "value_1_thread": round(band.size / one / 1e6, 3), "cpu": cpu_model(), "torch": torch.__version__,
            "sample": f"{reps} x one {w}x{h} Y frame on {cores} threads, float32 torch-CPU restatement of training/model.py "
                      f"model_blu, median {med:.3f} s/frame; 1 thread: one {band.shape[1]}x{band.shape[0]} band of it, {one:.2f} s"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    from qcnn_gpu_amd import capi, synth
    from qcnn_gpu_amd.formats import LAYERS, MACS_PER_PIXEL
    from qcnn_gpu_amd.qvrcnn import qvrcnn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # control plane only: barrier + max of the elapsed time
        # gloo announces its connections on stdout; the contract is ONE line there, so the rendezvous and
        # the first barrier run with file descriptor 1 pointed at stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    lib = capi.load()
    ndev = C.c_int()
    capi.check(lib.qcnn_device_count(C.byref(ndev)))
    if ndev.value < world and not a.allow_shared_gpus:
        raise SystemExit(f"bench.py: {world} ranks but only {ndev.value} GPU(s) present "
                         "(--allow-shared-gpus lets ranks share a device for a rehearsal)")
    dev = local_rank % max(ndev.value, 1)

    h, w, qp = a.height, a.width, a.qp
    hw = h * w
    B = a.batch if a.batch > 0 else max(16, ((1 << 30) // hw) // 16 * 16)
    m = synth.synth_model(qp)
    mb = m.to_nchw_vect_c_bytes()
    # distinct frames per rank: rank r cycles through its own four synthetic frames
    nsrc = min(B, 4)
    _, anc = synth.synth_frames(h, w, qp, frames=nsrc, seed=0x51434E4E + rank)
    net = qvrcnn(dev, 1, 1, h, w)
    if a.exact or a.wide:
        net.set_exact(1 if a.exact else 2)
    net.load_static_para(mb)
    fpl = min(net.batch_group(a.group), B)      # frames one launch of each fused kernel covers
    d_in, d_out = C.c_void_p(), C.c_void_p()
    capi.check(lib.qcnn_dev_alloc(C.byref(d_in), B * hw))
    capi.check(lib.qcnn_dev_alloc(C.byref(d_out), B * hw))
    for i in range(B):
        f = np.ascontiguousarray(anc[i % nsrc])
        capi.check(lib.qcnn_dev_h2d(C.c_void_p(d_in.value + i * hw), f.ctypes.data, hw))

    def step():
        net.forward_blu_batch(d_in.value, d_out.value, B, hw)

    def barrier():
        capi.check(lib.qcnn_dev_sync())
        if dist is not None:
            dist.barrier()

    # clock ramp: untimed launches until the device has been busy for --prewarm seconds
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < a.prewarm:
        net.forward_blu_batch(d_in.value, d_out.value, min(B, 4 * fpl), hw)
        capi.check(lib.qcnn_dev_sync())
    for _ in range(a.warmup):
        step()
    barrier()
    # ---- timed region: exactly --steps steps, no timing events on the stream ----
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    capi.check(lib.qcnn_dev_sync())
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())
    # ---- per-kernel launch times: HIP events on the engine stream, in an untimed pass of their own ----
    net.set_timing(True)
    net.kernel_times_us(reset=True)
    for _ in range(max(1, min(a.steps, 4))):
        step()
    ka_us, kb_us, passes = net.kernel_times_us(reset=True)
    net.set_timing(False)

    # the timed output must be the real thing: EVERY frame the timed steps wrote, whole, against the oracle
    # (outside the timed region; the batch cycles through nsrc distinct frames, so the oracle runs nsrc times
    # and every output frame is compared with the oracle's result for its source)
    parity = None
    e2e = None
    parity_frames = 0
    if True:
        from oracle import c_oracle
        parity = True
        rec_o = [c_oracle.forward_blu(mb, np.ascontiguousarray(anc[k])) for k in range(nsrc)]
        chunk = max(1, min(B, (256 << 20) // hw))
        buf = np.empty((chunk, h, w), np.uint8)
        for i0 in range(0, B, chunk):
            n = min(chunk, B - i0)
            capi.check(lib.qcnn_dev_d2h(buf.ctypes.data, C.c_void_p(d_out.value + i0 * hw), n * hw))
            for j in range(n):
                parity = parity and bool(np.array_equal(buf[j], rec_o[(i0 + j) % nsrc]))
            parity_frames += n
        del buf
    # what every rank saw, gathered on all ranks: a straggling or doubly used GPU shows in the one line
    mine = {"rank": rank, "device": dev, "pci_bus_id": capi.pci_bus_id(dev), "host": socket.gethostname(),
            "mpixels_s": round(a.steps * B * hw / elapsed_local / 1e6, 2), "parity": bool(parity),
            "k_c1c2_us": round(ka_us, 1), "k_c3c4_us": round(kb_us, 1)}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        parity = all(r["parity"] for r in ranks)
        parity_frames *= world
        places = [(r["host"], r["pci_bus_id"]) for r in ranks]
        if ndev.value >= world and len(set(places)) != world:
            raise SystemExit(f"bench.py: {world} ranks landed on {len(set(places))} distinct device(s): {places}")
    if rank == 0:
        rec = rec_o[0]
        if world == 1 and not a.no_e2e:
            host_in = np.ascontiguousarray(anc[0])
            host_out = np.empty_like(host_in)
            net.bind_io(0, 0)
            for _ in range(3):
                net.load_data(host_in); net.forward_blu(); net.download()
            n = 20
            t1 = time.perf_counter()
            for _ in range(n):
                net.load_data(host_in)     # blocking H2D   (inference/kernel.cu:93)
                net.forward_blu()          #                (inference/kernel.cu:94)
                host_out = net.download()  # sync + D2H     (inference/kernel.cu:95-96)
            e2e = n * hw / (time.perf_counter() - t1) / 1e6
            # the same region through the pipelined clip entry point (pinned DMA overlapping the kernels)
            # on page-locked host buffers allocated once, as a production caller would hold them
            nclip = 256
            p_in, p_out = C.c_void_p(), C.c_void_p()
            capi.check(lib.qcnn_host_alloc_pinned(C.byref(p_in), nclip * hw))
            capi.check(lib.qcnn_host_alloc_pinned(C.byref(p_out), nclip * hw))
            clip = np.ctypeslib.as_array(C.cast(p_in, C.POINTER(C.c_uint8)), shape=(nclip, h, w))
            rec_clip = np.ctypeslib.as_array(C.cast(p_out, C.POINTER(C.c_uint8)), shape=(nclip, h, w))
            for i in range(nclip):
                clip[i] = anc[i % nsrc]
            # (an untimed pass over the same clip first: filling the host buffers above left the device idle, and the
            # shader clock needs a few milliseconds to come back, as in the resident leg's untimed ramp)
            capi.check(lib.qcnn_forward_blu_clip(net._h, p_in, p_out, nclip, hw, None, None, None))
            t1 = time.perf_counter()
            capi.check(lib.qcnn_forward_blu_clip(net._h, p_in, p_out, nclip, hw, None, None, None))
            e2e_clip = nclip * hw / (time.perf_counter() - t1) / 1e6
            for i in range(nclip):
                parity = parity and bool(np.array_equal(rec_clip[i], rec_o[i % nsrc]))
            capi.check(lib.qcnn_host_free_pinned(p_in))
            capi.check(lib.qcnn_host_free_pinned(p_out))

    if rank == 0:
        pixels = float(world) * a.steps * B * hw
        value = pixels / elapsed / 1e6
        # dominant kernel: k_c1c2 (C1 + C2_1 + C2_2 = 45 632 of the 54 512 MAC/pixel)
        macs_a = sum(k * k * ci * co for _, k, ci, co in LAYERS[:3])
        ops_a = 2.0 * macs_a * hw * fpl
        achieved = ops_a / (ka_us * 1e-6) / 1e12 if ka_us > 0 else 0.0
        traffic = None
        hbm = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("height") == h and tj.get("width") == w:
                    # measured with frames_per_launch frames per dispatch; a dispatch of fpl frames moves fpl/that as much
                    scale = fpl / tj.get("frames_per_launch", 1)
                    traffic = round(tj["k_c1c2_hbm_bytes_per_launch"] * scale)
                    both = (tj["k_c1c2_hbm_bytes_per_launch"] + tj["k_c3c4_hbm_bytes_per_launch"]) * scale
                    if ka_us + kb_us > 0:
                        gbs = both / ((ka_us + kb_us) * 1e-6) / 1e9
                        hbm = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "bytes_per_pixel_moved": round(both / (hw * fpl), 1), "bytes_per_pixel_algorithmic": 2,
                               "note": "fabric-side bytes of both fused kernels (rocprofv3 FETCH_SIZE/WRITE_SIZE, profiles/traffic.json) "
                                       "over their summed launch time; the conc1 hand-off between the two kernels is what moves"}
            except Exception:
                traffic = None
        shape_tag = {(1080, 1920, 37): "configs[1]", (2160, 3840, 32): "configs[3] shape"}.get((h, w, qp), "custom")
        out = {
            "metric": "Mpixels/s, forward_blu on 1920x1080 Y frames, QP37 int8 QVRCNN" if (h, w, qp) == (1080, 1920, 37)
                      else f"Mpixels/s, forward_blu on {w}x{h} Y frames, QP{qp} int8 QVRCNN",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": f"{shape_tag}: {w}x{h} Y frame, QP{qp} int8, frames resident in HBM",
                       "frames_per_step_per_gpu": B, "frames_per_launch": fpl, "frame": [h, w], "qp": qp,
                       "epilogue": {(0, 0): "integer (proved at load)", (2, 0): "integer, C2 accumulators clamped first"}.get(
                           net.arith_paths, "float pipeline (EXACT)"),
                       "devices_present": ndev.value, "prewarm_s": a.prewarm,
                       "sharding": f"frames over {world} GPU(s), no collective"},
            "roofline": {"bound": "mfma", "kernel": "k_c1c2 (I1.ppro + C1 + C2_1 + C2_2 + concat)",
                         "achieved": round(achieved, 2), "peak": PEAK_INT8_TOPS, "unit": "TOP/s",
                         "frac": round(achieved / PEAK_INT8_TOPS, 4), "traffic": traffic,
                         "note": "int8 ops (2 per MAC); algorithmic 91 264 ops/pixel for this kernel x %d pixels per launch "
                                 "(%d frames); avg launch %.1f us (k_c3c4: %.1f us) over %d launches, HIP events on the engine "
                                 "stream in an untimed pass after the timed region; peak = nominal dense int8 rate at 2.4 GHz (a loop of "
                                 "nothing but int8 MFMAs on random operands sustains 3.9-4.1 POP/s at 1.9-2.0 GHz on this device class: "
                                 "profiles/r02_issue_ubench.txt)"
                                 % (hw * fpl, fpl, ka_us, kb_us, passes),
                         "whole_pass_frac": round((2.0 * MACS_PER_PIXEL * hw * fpl / ((ka_us + kb_us) * 1e-6) / 1e12) / PEAK_INT8_TOPS, 4)
                                            if ka_us + kb_us > 0 else None,
                         "whole_pass_frac_wall": round(2.0 * MACS_PER_PIXEL * value * 1e6 / world / 1e12 / PEAK_INT8_TOPS, 4)},
            "parity_vs_oracle": parity,
            "parity_frames_checked": parity_frames,
            "per_rank_mpixels_s": [r["mpixels_s"] for r in ranks],
            "per_rank_device": [f'{r["host"]}:{r["device"]}:{r["pci_bus_id"]}' for r in ranks],
        }
        if world > 1:
            out["per_rank_kernel_us"] = [[r["k_c1c2_us"], r["k_c3c4_us"]] for r in ranks]
        if ndev.value < world:
            out["config"]["note"] = f"{world} ranks share {ndev.value} device(s): a rehearsal, not a scaling measurement"
        if hbm is not None:
            out["roofline"]["hbm"] = hbm
        if e2e is not None:
            out["pcie_inclusive_mpixels_s"] = round(e2e, 2)
            out["pcie_inclusive_pipelined_mpixels_s"] = round(e2e_clip, 2)
        if not a.no_cpu_baseline and world == 1:   # the CPU baseline is timed at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(qp, h, w)
            except Exception as ex:  # the baseline is reported, never required for the GPU number
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    capi.check(lib.qcnn_dev_free(d_in))
    capi.check(lib.qcnn_dev_free(d_out))
    net.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
