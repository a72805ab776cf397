#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload m1|c2|c3|c4]

A *step* is one full render pass of the workload's frame through the C ABI (renderPassDevice ->
yafgpu_render_tiles, then the film combine; for N > 1 each rank renders its tile shard, combines its
own splat planes into one [H][W][5] film and the films are sum-reduced to rank 0 over RCCL).  A *ray* is one kd-tree query (closest-hit
or any-hit), counted by device atomics.  The scene is uploaded and its kd-tree built before the timed
region (reported separately); inputs are resident in HBM when timing starts.

Workloads:
  m1  1M-triangle diffuse Cornell-box-style soup, 512x512, 64 spp, primary + 1 bounce   (DEFAULT: the configuration
      BASELINE.json's `metric` is quoted on — "1M-tri scene @512^2 64spp")
  c2  100k-triangle diffuse Cornell box, 512x512, 64 spp, primary + 1 bounce            (configs[1])
  c3  1M-triangle diffuse soup, 1024x1024, 256 spp, 2 bounces                           (configs[2])
  c4  1M-triangle soup, 50% glossy, 2 area lights (BSDF sampling + MIS), 1024x1024, 64 spp  (configs[3])

Output: one JSON line with the contract fields plus `roofline` (dominant kernel: wf_trace; `achieved` =
SURVEY 8(d) algorithmic bytes / launch time measured live with HIP events; `traffic` and `hbm_measured_GBps`
= the memory-side bytes of the committed rocprofv3 PMC run of this workload) and `cpu_baseline` (the CPU
oracle — a port of the reference's algorithm — timed single-threaded and on all host cores of this box's
CPU share on a bounded sample of the same scene).
"""
import argparse
import hashlib
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    "c2": dict(desc="100k-tri diffuse Cornell box 512x512 64spp primary+1-bounce", n_tris=100_000, res=512, spp=64, bounces=1,
               glossy=0.0, lights=1, sigma=0.02, seed=1234),
    "c3": dict(desc="1M-tri diffuse soup 1024x1024 256spp 2 bounces", n_tris=1_000_000, res=1024, spp=256, bounces=2,
               glossy=0.0, lights=1, sigma=0.01, seed=1),
    "m1": dict(desc="1M-tri diffuse Cornell-box soup 512x512 64spp primary+1-bounce (BASELINE.json metric config)", n_tris=1_000_000, res=512, spp=64, bounces=1,
               glossy=0.0, lights=1, sigma=0.01, seed=1),
    "c4": dict(desc="1M-tri soup 50% glossy + 2 area lights (MIS) 1024x1024 64spp 2 bounces", n_tris=1_000_000, res=1024, spp=64,
               bounces=2, glossy=0.5, lights=2, sigma=0.01, seed=1),
}


KERNEL_SOURCES = ("yafgpu_wavefront.h", "yafgpu_device.hip", "yafgpu_shading.h", "yafgpu_math.h", "yafgpu_texture.h", "yafgpu_shade_variant.hip", "build.sh")


def kernel_key():
    """identifies the kernels a PMC summary was taken on: a hash over the device sources and the build flags (the GPU box has no .git)"""
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "libyafaray_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    h.update(os.environ.get("YAFGPU_EXTRA_FLAGS", "").encode())
    return h.hexdigest()[:16]


def find_traffic(workload):
    """the committed rocprofv3 PMC summary of this workload (profiles/<round>_<workload>_traffic.json, written by tools/pmc.sh):
    -> (doc, stale).  A summary whose `kernel_key` is not the built kernels' is stale: its byte counts are not quoted."""
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_traffic.json")))
    if not cands:
        return None, False
    key = kernel_key()
    docs = []
    for c in cands:
        try:
            d = json.load(open(c))
            d["_file"] = os.path.basename(c)
            docs.append(d)
        except Exception:
            pass
    for d in reversed(docs):
        if d.get("kernel_key") == key and d.get("workload", workload) == workload:
            return d, False
    return (docs[-1], True) if docs else (None, False)


def self_launch(args):
    """`python bench.py --gpus N` on its own: start the N ranks (one process per GPU) with torch.distributed.run as a CHILD, before
    anything in this process touches the GPU, hand its output through and leave with its exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def make_workload(name, res=None, spp=None, lights=None):
    from libyafaray_amd import scenes
    w = dict(WORKLOADS[name])
    if res:
        w["res"] = res
    if spp:
        w["spp"] = spp
    if lights:
        w["lights"] = lights      # C4's parity variant has one light (the two-light counter is serial state)
    sc = scenes.cornell_soup(w["n_tris"], seed=w["seed"], sigma=w["sigma"], glossy_fraction=w["glossy"], n_lights=w["lights"],
                             res=(w["res"], w["res"]))
    rd = scenes.render_settings(w["res"], w["res"], w["spp"], bounces=w["bounces"])
    return w, sc, rd


def host_cpu_share():
    """CPUs this process may actually use: the affinity mask, cut by a cgroup CPU quota when one is set (a one-GPU box
    shows all of the host's hardware threads in its affinity mask but owns a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                     # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    if quota is None:
        try:                                 # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(name, budget_s=18.0):
    """Time the CPU oracle (a port of the reference path) on a bounded sample of the workload: the same scene and spp
    at a reduced frame.  Three timings on the same sample: 1 thread, the box's CPU share (cgroup quota / affinity),
    and — where the affinity mask is wider than that — a thread per visible hardware thread, to show what
    oversubscription does.  `value` is the best multi-threaded rate; the 1-thread rate and the efficiency are beside it."""
    from oracle import pyoracle as po
    spp = WORKLOADS[name]["spp"]
    share = host_cpu_share()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else share
    # ONE oracle scene (its kd build for 1M triangles takes seconds) at twice the workload's camera resolution; every
    # timing renders a centred square window of that frame — same scene, same spp, a bounded number of pixels
    full = 2 * WORKLOADS[name]["res"]
    w, sc, rd = make_workload(name, res=full)
    t0 = time.time()
    osc = po.OracleScene(sc)                 # kd build is setup, not timed
    build_s = time.time() - t0

    def run(res, threads):
        res = min(res, full)
        o0 = (full - res) // 2
        _, s_ = osc.render(dict(rd, AA_minsamples=spp, oracle_threads=threads, tile_size=8, xstart=o0, ystart=o0, width=res, height=res))
        return s_.rays_closest + s_.rays_shadow, s_.render_seconds

    pilot_res = 32
    rays_p, sec_p = run(pilot_res, 1)
    rate1 = rays_p / max(sec_p, 1e-6)        # rays/s, pilot
    rays_per_px = rays_p / (pilot_res * pilot_res)

    # 1 thread: ~budget/4 seconds of work
    res1 = int(max(32, min(512, (rate1 * budget_s / 4.0 / rays_per_px) ** 0.5))) // 8 * 8
    rays_1, sec_1 = run(res1, 1)
    r1 = rays_1 / sec_1
    # multi-threaded: the CPU share, and when no quota could be read and the mask is wide (a one-GPU box shows all 256
    # hardware threads of the host), 16 and 64 threads as well — the table says where the scaling stops
    counts = sorted({share} | ({16, 64} if share > 64 else set()))
    per_run = budget_s * 0.6 / len(counts)
    table, best = {}, None
    for t in counts:
        rest = int(max(64, min(2048, (r1 * min(t, 32) * per_run / rays_per_px) ** 0.5))) // 8 * 8
        rays_t, sec_t = run(rest, t)
        table[str(t)] = {"Mrays_s": round(rays_t / sec_t / 1e6, 4), "px": rest, "rays": rays_t, "s": round(sec_t, 2),
                         "efficiency_vs_1_thread": round(rays_t / sec_t / (r1 * t), 3)}
        if best is None or rays_t / sec_t > best[1]:
            best = (t, rays_t / sec_t, rest, rays_t, sec_t)
    osc.close()
    return {"value": round(best[1] / 1e6, 4), "unit": "Mrays/s", "cores": best[0], "kind": "port",
            "one_thread_Mrays_s": round(r1 / 1e6, 4), "parallel_efficiency": round(best[1] / (r1 * best[0]), 3),
            "visible_hw_threads": visible, "cpu_share": share, "threads_table": table,
            "sample": f"same scene ({w['n_tris']} tris) at {full}x{full}, {spp} spp, centred windows: 1 thread {res1}x{res1} px, {rays_1} rays in {sec_1:.2f} s; "
                      f"{best[0]} threads {best[2]}x{best[2]} px, {best[3]} rays in {best[4]:.2f} s (oracle kd build {build_s:.1f} s excluded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="m1", choices=sorted(WORKLOADS))
    ap.add_argument("--res", type=int, default=0, help="override resolution (debug)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N-rank rehearsal on a ONE-GPU box: every rank uses cuda:0 and the film reduce goes through gloo "
                         "(host staging). Exercises sharding + reduce + combine with the real kernels; not a measurement.")
    ap.add_argument("--emulate-shard", type=int, default=0,
                    help="debug: render shard 0 of K on this one GPU (what one rank of a K-GPU run does per pass); not a measurement of K GPUs")
    ap.add_argument("--reduce", default="auto", choices=["auto", "rccl", "torch"],
                    help="film reduce of an N-rank run: 'rccl' = the C ABI's ncclReduce (yafaray_reduceFilm), 'torch' = torch.distributed.reduce, "
                         "'auto' = rccl, falling back to torch (and saying so) if the communicator cannot be created")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    from libyafaray_amd import Interface, scenes, interface as yi_mod
    from libyafaray_amd.parallel import reduce_film

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without it (bench.py starts its ranks itself)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w, sc, rd = make_workload(args.workload, args.res, args.spp)
    yi = Interface()
    scenes.load_scene(yi, sc, rd)
    yi.setShard(rank, world)            # pixel-tile sharding: tile t -> rank t % world
    comm, reduce_kind = None, "none"
    if world > 1:
        # the frame's collective: the C ABI's RCCL communicator (yafaray_reduceFilm = ncclReduce over xGMI, csrc/yafaray_reduce.cpp);
        # torch.distributed only carried the 128-byte id and keeps the barrier / timing reductions.  Every rank takes the same branch.
        reduce_kind = "torch.distributed"
        if not args.rehearse_one_gpu and args.reduce in ("auto", "rccl"):
            from libyafaray_amd.parallel import FilmComm
            ok = torch.ones(1, device=dev)
            try:
                comm = FilmComm.from_process_group(local_rank)
            except Exception as e:      # noqa: BLE001
                if args.reduce == "rccl":
                    raise
                print(f"[bench] rank {rank}: C-ABI RCCL communicator failed ({e}); falling back to torch.distributed.reduce", file=sys.stderr)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() > 0:
                reduce_kind = "rccl (C ABI yafaray_reduceFilm; " + comm.backend + ")"
            elif comm is not None:
                comm.close()
                comm = None
        # only workloads that consume the reference's serial light counter (several lights: c4) ever call the exchange: a few KB of
        # per-tile call counts per pass, so that the ranks make the single-GPU render's light choices
        if comm is not None:
            yi.setComm(comm)
        else:
            from libyafaray_amd.parallel import plane_exchange
            yi.setPlaneExchange(plane_exchange(dev))
    if args.emulate_shard > 1 and world == 1:
        yi.setShard(0, args.emulate_shard)
    t0 = time.time()
    yi.prepareRender()                  # Scene::update: kd-tree build + upload (untimed, reported)
    setup_s = time.time() - t0
    stats0 = yi.getRenderStats()
    W, H = yi.getRenderSize()

    planes = torch.zeros((4, H, W, 5), dtype=torch.float32, device=dev)
    film = torch.zeros((H, W, 5), dtype=torch.float32, device=dev)
    counters = torch.zeros(8, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def reduce():
        # one combined [H][W][5] film per rank (5 MB at 512^2), not the four splat planes: sum over ranks of per-rank
        # combines == combine of the summed planes up to the order of <= 2 additions on tile-border pixels
        if args.rehearse_one_gpu and world > 1:
            host = film.cpu()
            reduce_film(host, dst=0)
            film.copy_(host)
        elif comm is not None:
            comm.reduce_film(film, dst=0, stream=stream)
        else:
            reduce_film(film, dst=0)

    def step():
        yi.renderPassDevice(planes.data_ptr(), counters.data_ptr(), stream)
        yi_mod.film_combine(planes.data_ptr(), film.data_ptr(), W, H, stream)
        reduce()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    counters.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        yi.renderPassDevice(planes.data_ptr(), counters.data_ptr(), stream)
        ev[k][1].record()
        yi_mod.film_combine(planes.data_ptr(), film.data_ptr(), W, H, stream)
        reduce()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # render_kernel (+ its memset/tile upload), same stream

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    cnt = counters.clone()
    if args.rehearse_one_gpu:
        el, cnt = el.cpu(), cnt.cpu()
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    c = cnt.cpu().numpy()
    rays_total = int(c[0] + c[1])
    value = rays_total / elapsed / 1e6

    # per-ray traversal averages for the algorithmic-bytes figure + per-kernel device time: one extra,
    # untimed pass of the counting variant of the traversal kernels with HIP events around every launch
    os.environ["YAFGPU_STATS"] = "1"
    counters.zero_()
    yi.setProfiling(True)
    yi.renderPassDevice(planes.data_ptr(), counters.data_ptr(), stream)
    torch.cuda.synchronize()
    prof_stats = yi.getKernelProfile()
    del os.environ["YAFGPU_STATS"]
    s = counters.cpu().numpy().astype(np.float64)
    # ... and once more with the production kernels for the durations that go into the roofline
    counters.zero_()
    yi.renderPassDevice(planes.data_ptr(), counters.data_ptr(), stream)
    torch.cuda.synchronize()
    prof = yi.getKernelProfile()
    yi.setProfiling(False)
    rays_launch = s[0] + s[1]
    n_int, n_leaf, n_tri = s[2] / rays_launch, s[3] / rays_launch, s[4] / rays_launch
    # bytes/ray = 32 (ray) + 16 (hit) + 8*(interior+leaf nodes) + 4*leaf refs + 48*triangle records
    bytes_per_ray = 32 + 16 + 8 * (n_int + n_leaf) + 4 * n_tri + 48 * n_tri
    trace_ms = prof["trace_closest"][0] + prof["trace_shadow"][0]
    trace_launches = prof["trace_closest"][1] + prof["trace_shadow"][1]
    if trace_launches == 0:       # one-kernel pipeline (YAFGPU_PIPELINE=megakernel): the pass is the kernel
        trace_ms, trace_launches = kernel_ms, 1
    # dominant kernel = the traversal kernel (wf_trace): algorithmic bytes of all its launches / their summed duration
    achieved = bytes_per_ray * rays_launch / (trace_ms * 1e-3) / 1e9
    # HBM-side bytes per launch of the same kernel come from a rocprofv3 PMC run (FETCH_SIZE / WRITE_SIZE in separate
    # passes, tools/pmc.sh); they cannot be read live, so the committed summary of that run of THIS workload is quoted
    # when it exists (profiles/r02_<workload>_traffic.json; it names its own correction of FETCH_SIZE)
    traffic, pmc, pmc_stale = None, None, False
    if world == 1 and not args.res and not args.spp:
        pmc, pmc_stale = find_traffic(args.workload)
        if pmc is not None and not pmc_stale:
            traffic = round(pmc["traffic_bytes_per_launch"])
    avg_launch_s = trace_ms * 1e-3 / max(trace_launches, 1)
    hbm_measured = (traffic / avg_launch_s / 1e9) if traffic else None

    if rank == 0:
        out = {
            "metric": "Mrays/sec (primary+secondary) on the path-tracing hot path", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["desc"], "triangles": int(stats0.n_triangles), "width": W, "height": H, "spp": w["spp"],
                       "bounces": w["bounces"], "lights": w["lights"], "parallelism": f"pixel-tile shard x{world}", "film_reduce": reduce_kind,
                       "rays_per_step": rays_total // args.steps, "rays_per_camera_sample": round(rays_total / max(int(c[5]), 1), 3),
                       "kd_nodes": int(stats0.kd_nodes), "kd_leaf_refs": int(stats0.kd_leaf_refs), "kd_max_depth": int(stats0.kd_max_depth),
                       "scene_device_MB": round(stats0.scene_device_bytes / 1e6, 1), "setup_s": round(setup_s, 2),
                       "tree_build_s": round(stats0.tree_build_seconds, 2)},
            # `bound` names what limits the kernel (PMC: VALU issue slots ~0.8 busy at < 0.5 lane utilisation, working set cache-resident,
            # counter-side HBM rate 0.2-0.4 of peak); `achieved` / `peak` / `frac` are nevertheless the bench contract's accounting
            # against the HBM roofline (`accounting`), with the counter-side figure beside them (`hbm_measured_*`)
            "roofline": {"bound": "valu-issue (cache-resident)", "accounting": "hbm: SURVEY 8(d) algorithmic bytes / measured launch time over the 8 TB/s HBM peak",
                         "kernel": "wf_trace (closest-hit + any-hit kd traversal)", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "pmc_stale": bool(pmc_stale), "kernel_key": kernel_key(),
                         "pmc_file": None if pmc is None else pmc.get("_file"),
                         # `achieved` / `frac` follow the bench contract: SURVEY 8(d) ALGORITHMIC bytes / measured launch time.
                         # The memory-side figure is beside it: counter bytes (traffic) / the same launch time.  The scene
                         # (~100 MB at 1M triangles) is L2 / Infinity-Cache resident, so the kernel is bound by vector
                         # instruction issue under divergence, not by HBM (see `limiter`).
                         "algorithmic_GBps": round(achieved, 2),
                         "hbm_measured_GBps": None if hbm_measured is None else round(hbm_measured, 2),
                         "hbm_measured_frac": None if hbm_measured is None else round(hbm_measured / HBM_PEAK_GBS, 5),
                         "limiter": "VALU issue under SIMT divergence (working set cache-resident)",
                         "pmc": None if (pmc is None or pmc_stale) else {k: pmc[k] for k in ("source", "fetch_correction", "fetch_bytes_per_launch_raw",
                                                                            "write_bytes_per_launch", "lane_utilisation", "salu_per_valu",
                                                                            "shade") if k in pmc},
                         "algorithmic_bytes_per_launch": round(bytes_per_ray * rays_launch / max(trace_launches, 1)),
                         "launches_per_pass": int(trace_launches), "avg_launch_ms": round(trace_ms / max(trace_launches, 1), 4),
                         "rays_per_launch": round(rays_launch / max(trace_launches, 1)),
                         "pass_ms": {k: round(v[0], 3) for k, v in prof.items()}, "pass_ms_total": round(kernel_ms, 3),
                         "bytes_per_ray": round(bytes_per_ray, 1),
                         "per_ray": {"interior_nodes": round(n_int, 2), "leaves": round(n_leaf, 2), "tri_tests": round(n_tri, 2),
                                     "restarts": round(s[6] / rays_launch, 5)},
                         # lanes doing a step per wave round of that kind / 64 (stats pass only)
                         "simt": {"node_rounds": round((int(s[2]) + int(s[3])) / max(64 * (int(s[7]) & 0xffffffff), 1), 3),
                                  "tri_rounds": round(int(s[4]) / max(64 * (int(s[7]) >> 32), 1), 3)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        if args.rehearse_one_gpu:
            f = film.cpu().numpy()
            out["film_checksum"] = {"sum_rgb": float(f[..., :3].astype(np.float64).sum()), "sum_weight": float(f[..., 4].astype(np.float64).sum()),
                                    "crc": int(np.frombuffer(f.tobytes(), dtype=np.uint32).astype(np.uint64).sum() & 0xffffffff)}
        print(json.dumps(out))
    if comm is not None:
        yi.setComm(None)
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
