#!/usr/bin/env python3
"""bench.py -- the entity transform + cull hot path on N MI355X of one node.

A "step" is one frame of BASELINE.json configs[1] on every GPU: 1M entities in a
depth-8 transform hierarchy (125k chains x 8, synthetic), every entity
dirty, resident in HBM: TRS -> world matrix -> inverse -> world AABB -> frustum cull
-> ascending visible-index list.  With N > 1 the scene is ONE global forest of N x 125k
chains sharded by entity range: rank r owns block r (whole subtrees; the blocks are the
tile ranges clapgpu_shard_tile_range cuts, tests/test_shard_cpu.py), global id = r * n_pad +
local slot; weak scaling, no data-path collective, and the only exchange is the RCCL
allgather of the compacted visible set (clapgpu_exchange_visible, C).  --c5 = BASELINE
configs[4]'s sizing (2 M entities + 256 k particles per GPU: 16 M + 2 M at 8 GPUs).

Started plainly with --gpus N > 1 (no torch.distributed.run environment) it is its own launcher: N fresh child
processes, one per GPU, before this process makes any GPU call (launch_ranks).

Prints ONE JSON line (rank 0).  `value` = entity updates / s over all GPUs.
`roofline` = algorithmic bytes of the update kernel / its launch-to-launch time, K launches back to
back between one HIP event pair on the stream they are launched on (a separate pass after the timed
region, same process, same data); `step_us` and `expand_launch_us` beside it make the line checkable
against itself.  `cpu_baseline` = the reference's own default_update + view_entity_in_frustum
(oracle/_ref, built from the reference sources) on one host core, or the oracle port if that binary
is absent.
"""
import argparse
import glob
import json
import os
import sys
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# key order of the ONE line (short scalars first, the long sections last: a record that keeps the head of the line keeps
# `summary` and `c5`), and the keys of the second leg an N > 1 run times beside the headline (BASELINE configs[4]'s share)
LINE_ORDER = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "rccl_ranks", "devices", "ms_per_step_per_rank", "summary", "c5", "config", "roofline",
              "cpu_baseline", "extra"]
C5_KEYS = ["workload", "value", "unit", "particle_updates_per_s", "ms_per_step", "ms_per_step_per_rank", "steps", "warmup",
           "entities_per_gpu", "particles_per_gpu", "visible", "global_ids", "scaling"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chains", type=int, default=125_000, help="hierarchy chains per GPU (x depth = entities)")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--layout", choices=["tiles", "levels"], default="tiles",
                    help="tiles: subtree tiles, one launch for all levels; levels: level-major, one launch per level")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary workloads (pose/skinning, particles, bodies) reported under 'extra'")
    ap.add_argument("--particles", type=int, default=0,
                    help="also update this many particles per GPU (whole 1024-particle systems) inside every step: "
                         "BASELINE configs[4] is --gpus 8 --chains 250000 --particles 262144")
    ap.add_argument("--snapshot", default=None,
                    help="run the entity step on a scene snapshot (include/clapgpu_snapshot.h; components "
                         "'entities' and optionally 'camera') instead of the synthetic BASELINE workload")
    ap.add_argument("--c5", action="store_true",
                    help="BASELINE configs[4] sizing per GPU: 250 000 chains x depth 8 (2 M entities) + 262 144 particles, i.e. "
                         "16 M entities + 2 M particles at --gpus 8.  (The default at any N keeps configs[1]'s 1 M entities per GPU, "
                         "so that the driver's 1/2/4/8 curve is weak scaling of one workload.)")
    ap.add_argument("--exchange", choices=["rccl", "c10d"], default="rccl",
                    help="N > 1: call ncclAllGather directly (low host overhead) or through torch.distributed")
    ap.add_argument("--no-testbed", action="store_true",
                    help="skip the testbed-sized (BASELINE configs[0]) extra: under rocprofv3 its small launches of "
                         "the same kernels would be averaged into the full-size per-kernel statistics")
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--init-timeout", type=float, default=300.0,
                    help="N > 1: seconds a rank may spend in the collective set-up (process group + ncclCommInitRank) before it "
                         "gives up with exit code 4 -- a rank stuck there would hang the launcher and every other rank")
    ap.add_argument("--dry-launch", action="store_true",
                    help="self-test of the --gpus N launcher: the ranks rendezvous over gloo on the CPU, reduce their ranks and "
                         "rank 0 prints one JSON line; no GPU is touched (tests/test_bench_launcher.py)")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`bench.py --gpus N` started plainly (no torch.distributed.run environment): start the N ranks ourselves, one
    fresh process per GPU, BEFORE this process makes any GPU call -- a process that has initialised the GPU must never
    be replaced or forked into ranks.  Rank 0's stdout (the ONE JSON line) is passed through; the other ranks' stdout
    goes to stderr.  Returns the exit code: 0 only if every rank exited 0; a failing rank takes the others down (exact
    PIDs, never a pattern)."""
    n = args.gpus
    if not args.dry_launch:
        # the node's GPUs without touching HIP in this parent (torch.cuda.device_count() falls back to hipGetDeviceCount,
        # which initialises the runtime, where amdsmi is missing): KFD's topology lists one node per agent, GPUs have SIMDs
        have = 0
        for node in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
            try:
                with open(node) as fh:
                    if any(l.startswith("simd_count") and int(l.split()[1]) > 0 for l in fh):
                        have += 1
            except OSError:
                pass
        vis = os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES")
        if vis is not None and vis.strip() != "":
            have = min(have, len([v for v in vis.split(",") if v.strip() != ""])) if have else len(vis.split(","))
        if have == 0 and os.path.isdir("/sys/class/kfd/kfd/topology/nodes") and not os.access("/sys/class/kfd/kfd/topology/nodes", os.R_OK | os.X_OK):
            import torch                                     # the topology exists but cannot be read here: ask the runtime after all
            have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this pool
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=(None if r == 0 else sys.stderr)))
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        if live:
            time.sleep(0.05)
    return rc


class InitWatchdog:
    """A rank that does not get through its collective set-up within `seconds` ends ITSELF with exit code 4 (the launcher --
    ours or torch.distributed.run -- then stops the others): ncclCommInitRank and the process group's rendezvous block
    for ever when a peer never arrives.  A fresh thread, os._exit: nothing of a wedged runtime is waited for."""

    def __init__(self, seconds, what):
        import threading
        self._t = threading.Timer(seconds, self._fire, args=(seconds, what))
        self._t.daemon = True
        self._t.start()

    @staticmethod
    def _fire(seconds, what):
        print(f"[bench] rank {os.environ.get('RANK', '0')}: {what} did not finish within {seconds:.0f} s -- giving up",
              file=sys.stderr, flush=True)
        os._exit(4)

    def done(self):
        self._t.cancel()


def check_proof(proof, gpus):
    """The run's own evidence that `gpus` ranks on `gpus` distinct devices took part; returns an error string or None."""
    if proof["rccl_ranks"] != gpus:
        return f"the communicator has {proof['rccl_ranks']} ranks, --gpus is {gpus}"
    if not proof["comm_rank_ok"]:
        return "a rank's communicator rank is not its launcher rank"
    if len(proof["devices"]) != gpus or len(set(proof["devices"])) != gpus:
        return f"{gpus} ranks on {len(set(proof['devices']))} distinct device(s): {proof['devices']}"
    return None


def dry_launch():
    """The launcher's self-test body (one rank): gloo rendezvous on the CPU with the environment launch_ranks() made, the
    same proof-of-participation check the GPU run ends on (devices stand-ins: "cpu:<rank>")."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("CLAP_BENCH_DRY_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    wd = InitWatchdog(float(os.environ.get("CLAP_BENCH_DRY_INIT_TIMEOUT", "120")), "the gloo rendezvous")
    if os.environ.get("CLAP_BENCH_DRY_HANG_RANK") == str(rank):
        time.sleep(3600)                                     # a rank that never arrives: its own watchdog ends it
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wd.done()
    t = torch.tensor([rank, int(os.environ["LOCAL_RANK"]), 1], dtype=torch.int64)
    dist.all_reduce(t)
    every = [None] * world
    same = os.environ.get("CLAP_BENCH_DRY_SAME_DEVICE") == "1"
    dist.all_gather_object(every, dict(rccl_ranks=dist.get_world_size(), comm_rank=dist.get_rank(),
                                       device="cpu:0" if same else f"cpu:{rank}"))
    proof = dict(rccl_ranks=min(e["rccl_ranks"] for e in every), comm_rank_ok=all(e["comm_rank"] == r for r, e in enumerate(every)),
                 devices=[e["device"] for e in every])
    bad = check_proof(proof, world)
    if bad:
        if rank == 0:
            print(f"[bench] refusing to report: {bad}", file=sys.stderr, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(5)
    if rank == 0:
        args = parse()
        two_legs = world > 1 and args.chains == 125_000 and args.depth == 8 and args.particles == 0 and not args.c5 and not args.snapshot
        print(json.dumps({"launcher": "dry", "n_gpus": world, "rank_sum": int(t[0]), "local_rank_sum": int(t[1]),
                          "ranks": int(t[2]), "rccl_ranks": proof["rccl_ranks"], "devices": proof["devices"],
                          # the shape of the line the GPU run of the same command prints: headline = configs[1] per GPU
                          # (weak scaling), and with N > 1 and no explicit sizing the configs[4] share as a second leg
                          "line_keys": [k for k in LINE_ORDER if k not in ("extra", "cpu_baseline") and (two_legs or k != "c5")],
                          "c5_keys": C5_KEYS if two_legs else None}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def cpu_all_cores(scene, cam, frames=8):
    """Context for the 1-thread baseline: the port's same frame on every host core (OpenMP over tiles)."""
    from oracle import binding as ob
    if "tile_row_start" not in scene:
        return None
    fr, _v, _p = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    mask = np.zeros((int(scene["n"]) + 63) // 64, np.uint64)
    t = []
    for _ in range(frames + 1):
        st["flags"] |= np.where(st["flags"] & 0x80000000, 1 << 16, 0).astype(np.uint32)
        t0 = time.perf_counter()
        ob.entities_frame_tiles_mt(scene, st, fr, mask)
        t.append(time.perf_counter() - t0)
    t = t[1:]                                                # first call spins up the thread pool
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    threads = ob.omp_max_threads()
    quota = cgroup_cpu_limit()
    # what the figure really ran on: OpenMP's thread count, capped by a container CPU quota when there is one
    effective = threads if quota is None else max(1, min(threads, int(np.ceil(quota))))
    return dict(value=int(scene["n_real"]) / (sum(t) / len(t)), unit="entity updates/s", cores=effective,
                omp_threads=threads, affinity_cores=affinity, cgroup_cpu_limit=quota,
                kind="port", sample=f"{frames} frames, oracle/ C restatement, OpenMP over tiles: {threads} threads "
                                    f"(omp_get_max_threads) on {affinity} schedulable cores, container CPU quota "
                                    f"{'none' if quota is None else f'{quota:.1f} cores'}")


def cgroup_cpu_limit():
    """CPU quota of this container in cores (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def dropin_boundary():
    """The reference's own objects on both sides (oracle/_ref/clap_dropin, tests/test_dropin.py): its
    mq_update + view_entity_in_frustum against the CLAP-side binding over libclapgpu_scene, host entity3d
    structs in, host entity3d structs out -- the PCIe- and scatter-inclusive cost of the boundary."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "clap_dropin")
    if not os.access(exe, os.X_OK):
        return None
    out = {}
    fields = ("reference_ms_per_frame", "binding_ms_per_frame", "reference_mq_update_ms", "binding_mq_update_ms", "binding_ms",
              "reference_mutate_ms", "binding_mutate_ms", "reference_render_block_ms", "binding_render_block_ms",
              "binding_draw_list_ms", "reference_frame_ms", "binding_frame_block_ms", "binding_frame_draw_list_ms",
              "drawn_per_frame", "left_stale_per_frame", "fetched_on_view_per_frame")

    def one(n, frames, permille, *policy):
        p = subprocess.run([exe, "bench", str(n), str(frames), str(permille), "notify", *policy], capture_output=True, text=True,
                           timeout=240)
        r = json.loads(p.stdout.strip().splitlines()[-1])
        d = {k: r[k] for k in fields}
        d["identical"] = (r["mismatches"] == 0 and r["visible_equal"] and r["draw_sets_equal"] and r["draw_reads_equal"])
        return d

    for n, frames, permille in ((10_000, 50, 1000), (10_000, 50, 100), (1_000_000, 5, 1000)):
        key = f"{n}_entities_{permille // 10}pct_dirty"
        try:
            # the default policy (every rebuilt entity3d written back) and GPU_SCATTER_DRAWN (what is read is written back)
            out[key] = dict(one(n, frames, permille), policy="GPU_SCATTER_ALL")
            out[key]["scatter_drawn"] = dict(one(n, frames, permille, "drawn"), policy="GPU_SCATTER_DRAWN")
        except Exception as e:                                   # a figure for context: never fail the bench on it
            out.setdefault(key, {})["error"] = repr(e)[:200]
    # a queue whose make-up changes EVERY frame (10 entities made -- roots, children of random earlier entities, light carriers --
    # and 10 leaves deleted; 10 % of the entities moving): gpu_scene_entity_created / _deleting place them into / take them out of
    # the standing device layout; before round 5 each such frame walked the queue and re-tiled it (1 M: 257-267 ms)
    for n, frames in ((10_000, 60), (1_000_000, 6)):
        key = f"{n}_entities_10pct_dirty_10_made_10_deleted_a_frame"
        try:
            p = subprocess.run([exe, "bench", str(n), str(frames), "100", "notify", "drawn", "churn", "10"], capture_output=True, text=True,
                               timeout=240)
            r = json.loads(p.stdout.strip().splitlines()[-1])
            out[key] = {k: r[k] for k in ("reference_mq_update_ms", "binding_mq_update_ms", "binding_ms", "reference_frame_ms",
                                          "binding_frame_draw_list_ms", "fast_frames", "retiles", "placed_in_layout", "removed_in_place")}
            out[key]["identical"] = (r["mismatches"] == 0 and r["visible_equal"] and r["draw_sets_equal"] and r["draw_reads_equal"])
            out[key]["policy"] = "GPU_SCATTER_DRAWN"
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    # a frame as pipeline_render renders it: two shadow passes with the LIGHT's view (registered: gpu_scene_add_view; culled by the
    # update's own launch) and no camera, then the model pass with the camera's (pipeline-builder.c:34-46, 246-272; model.c:752-760)
    for n, frames in ((10_000, 50), (1_000_000, 5)):
        key = f"{n}_entities_10pct_dirty_pipeline_frame_2_shadow_passes"
        try:
            p = subprocess.run([exe, "bench", str(n), str(frames), "100", "notify", "drawn", "shadow", "2"], capture_output=True, text=True,
                               timeout=300)
            r = json.loads(p.stdout.strip().splitlines()[-1])
            out[key] = {k: r[k] for k in ("reference_pipeline_frame_ms", "binding_pipeline_frame_block_ms", "binding_pipeline_frame_draw_list_ms",
                                          "reference_shadow_passes_ms", "binding_shadow_passes_ms", "binding_shadow_draw_lists_ms",
                                          "reference_mq_update_ms", "binding_mq_update_ms", "views_culled_per_update",
                                          "cull_launches_after_update", "shadow_drawn_per_frame", "drawn_per_frame")}
            out[key]["identical"] = (r["mismatches"] == 0 and r["visible_equal"] and r["draw_sets_equal"] and r["draw_reads_equal"] and
                                     r["shadow_sets_equal"])
            out[key]["policy"] = "GPU_SCATTER_DRAWN"
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    # NO notifications (the minimal patch): the queue is verified against the last walk's records on the workers and the frame
    # goes by the records; round 4 walked every entity3d twice on one core (1 M: 111 ms)
    try:
        p = subprocess.run([exe, "bench", "1000000", "5", "100"], capture_output=True, text=True, timeout=240)
        r = json.loads(p.stdout.strip().splitlines()[-1])
        out["1000000_entities_10pct_dirty_no_notifications"] = {
            **{k: r[k] for k in ("reference_mq_update_ms", "binding_mq_update_ms", "binding_ms", "reference_frame_ms",
                                 "binding_frame_draw_list_ms", "frames_by_the_records", "retiles")},
            "identical": (r["mismatches"] == 0 and r["visible_equal"] and r["draw_sets_equal"] and r["draw_reads_equal"])}
    except Exception as e:
        out["1000000_entities_10pct_dirty_no_notifications"] = {"error": repr(e)[:200]}
    # the two frame kinds that WALK the queue, at 1 M entities: every frame walked and re-tiled (GPU_SCENE_INCREMENTAL=0: what each
    # frame with an edit cost before round 5; round 5: 200-214 ms, the one frame kind slower than the reference), and every frame
    # walked with the layout standing (no notifications, GPU_SCENE_REPLAY=0).  The list chase on one thread, the rest on the workers
    for key, env, args in (("1000000_entities_10pct_dirty_walked_and_retiled_every_frame", {"GPU_SCENE_INCREMENTAL": "0"},
                            ["1000000", "5", "100", "notify", "churn", "10"]),
                           ("1000000_entities_10pct_dirty_walked_every_frame", {"GPU_SCENE_REPLAY": "0"}, ["1000000", "5", "100"])):
        try:
            p = subprocess.run([exe, "bench", *args], capture_output=True, text=True, timeout=240, env=dict(os.environ, **env))
            r = json.loads(p.stdout.strip().splitlines()[-1])
            out[key] = {**{k: r[k] for k in ("reference_mq_update_ms", "binding_mq_update_ms", "binding_ms", "fast_frames", "retiles",
                                             "frames_by_the_records")},
                        "identical": (r["mismatches"] == 0 and r["visible_equal"] and r["draw_sets_equal"] and r["draw_reads_equal"])}
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    # skeletal animation through the same boundary: animated_update per character on the host (clock, queue, channels_transform,
    # one_joint_transform: core/model.c:1266-1404, 1563-1591) against gpu_mq_update + gpu_anim_update
    # (10 x 64 over 400 frames: the testbed's own scale, core/clap.c's demo scenes -- three device round trips a frame)
    for chars, joints, frames in ((10, 64, 400), (500, 64, 14), (5_000, 64, 8)):
        key = f"{chars}_characters_{joints}_joints"
        try:
            p = subprocess.run([exe, "anim", str(chars), str(joints), str(frames), "5", "notify"], capture_output=True, text=True,
                               timeout=240)
            r = json.loads(p.stdout.strip().splitlines()[-1])
            out[key] = {"reference_ms_per_frame": r["reference_ms_per_frame"], "binding_ms_per_frame": r["binding_ms_per_frame"],
                        "worst_relative_error": r["worst_relative_error"], "differing_objects": r.get("differing_objects"),
                        "equal_to_the_reference": r["mismatches"] == 0 and r.get("differing_objects") == 0}
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    # particle systems: one particles_update hook per system (core/particle.c:89-140) against gpu_particles_update
    for systems, per, frames in ((20, 512, 400), (48, 1024, 32), (1024, 1024, 8)):
        key = f"{systems}_particle_systems_x_{per}"
        try:
            p = subprocess.run([exe, "particles", str(systems), str(per), str(frames), "4"], capture_output=True, text=True, timeout=240)
            r = json.loads(p.stdout.strip().splitlines()[-1])
            out[key] = {"reference_ms_per_frame": r["reference_ms_per_frame"],
                        "binding_ms_per_frame": r["binding_ms_per_frame_positions_only"],
                        "binding_ms_per_frame_with_particle_structs": r["binding_ms_per_frame_with_particle_structs"],
                        "identical": r["mismatches"] == 0 and r["stream_draws_agree"]}
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    out["note"] = ("random forest (60 % of the entities parented, parents listed before their children), camera drifting; *_ms_per_frame = "
                   "mq_update + one frustum verdict per entity asked in list order like _models_render (the caller's walk of the lists is "
                   "inside both sides), *_mq_update_ms = the update call alone, *_mutate_ms = the frame's entity3d_move calls (the binding's "
                   "carry the notification), *_render_block_ms = _models_render's per-entity block (model.c:958-992: verdict, LOD pick, the "
                   "draw's reads of mx / inverse_mx) over every entity, binding_draw_list_ms = gpu_scene_select_lod + the same reads over "
                   "gpu_scene_visible_of() per txmodel, *_frame_* = mutate + mq_update + that consumer.  Binding through the engine's own "
                   "names (mq_update, view_entity_in_frustum, entity3d_move ...), notification mode: touched transforms up, kernel, results "
                   "down, scattered back into the entity3d structs -- every rebuilt entity (GPU_SCATTER_ALL, the default) or what is read: "
                   "drawn / containing the camera / standing host readers, the rest fetched when it comes into view or on demand "
                   "(scatter_drawn: GPU_SCATTER_DRAWN); worker threads (<= 16) for frames that touch > 64 k entities.  "
                   "*_characters_*: the frame's mq_update with every character's animated_update (keyframes, hierarchy, palette, joint "
                   "positions) on the host against the binding (entities, pose on the device, T/R/S + palette + positions of every "
                   "joint copied back into the entity3d structs, joint-attached props in a second launch); every float the reference's, bit for bit.  "
                   "*_particle_systems_*: binding_ms_per_frame = what a frame needs (every system's pos_array and billboard matrix back "
                   "on the host, libc's drand48 position handed on), *_with_particle_structs = every struct particle's pos / velocity "
                   "written back as well (only when the game reads them)")
    return out


def cpu_baseline(scene, cam, frames):
    """Reported baseline, not the target: the reference (or the port) on one host core."""
    from oracle import binding as ob, refrun
    if "OMP_NUM_THREADS" not in os.environ:                 # as many threads as the container may actually run at once (libgomp
        quota = cgroup_cpu_limit()                          # read the environment when torch loaded it: set the team size directly)
        aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ob.omp_set_threads(max(1, min(aff, int(np.ceil(quota)) if quota else aff)))
    n_real = int(scene["n_real"])
    cores = 1
    if refrun.available():
        r = refrun.bench_entities(scene, cam, reps=frames)
        # the boundary figure first: the all-cores leg leaves an OpenMP team behind, and a team wider than the container's
        # CPU quota gets the whole cgroup throttled for the periods that follow
        boundary = dropin_boundary()
        return dict(value=n_real / r["mean_s"], unit="entity updates/s", cores=cores, kind="reference",
                    all_cores=cpu_all_cores(scene, cam), dropin_boundary=boundary,
                    sample=f"{frames} frames of the same 1-GPU workload ({n_real} entities, all dirty): the reference's "
                           "default_update + view_entity_in_frustum (core/model.c, core/view.c; ROCm clang -O2 "
                           f"-ffp-contract=off), 1 thread; best frame {n_real / r['best_s']:.3e}/s")
    fr, _v, _p = ob.frustum_from_camera(cam)
    st = ob.entity_state(scene)
    t = []
    for _ in range(frames):
        st["flags"] |= np.where(st["flags"] & 0x80000000, 1 << 16, 0).astype(np.uint32)
        t0 = time.perf_counter()
        ob.entities_update(scene, st)
        ob.entities_cull(scene["n"], st["flags"], st["aabb"], fr)
        t.append(time.perf_counter() - t0)
    return dict(value=n_real / (sum(t) / len(t)), unit="entity updates/s", cores=cores, kind="port",
                all_cores=cpu_all_cores(scene, cam),
                sample=f"{frames} frames of the same 1-GPU workload ({n_real} entities, all dirty), oracle/ C "
                       "restatement (gcc -O2 -ffp-contract=off), 1 thread")


from bench_extras import (HBM_PEAK_GBS, time_launches, spawn_cached, pmc_kernel_traffic, roof, snapshot_characters,   # noqa: E402,F401
                          extras, testbed_frame, full_frame, summary, secondary)


class RankStep:
    """One rank's share of the workload and its step, exactly what the timed loop runs (tests/test_c5_gpu.py drives this same
    object at BASELINE configs[4]'s per-rank size against the oracle).  Rank r owns block r of ONE global forest
    (synth.entities_chains seeded by its global chain numbers), global entity id = r * n_pad + local slot; particle
    systems shard by whole system with the rank's own drand48 stream.  With `use_dist` the frame's visible set leaves
    through the path's only exchange: the 1-bit-per-entity mask in ONE fixed-size RCCL allgather (no counts, no host
    sync), expanded on every rank into the identical ascending global id list; exchange + expansion of frame f run on a
    side stream under the update of frame f + 1 (two mask buffers)."""

    def __init__(self, chains, depth, particles, rank, world, device, use_dist, route="rccl", layout="tiles", raw=None,
                 cam=None, block=None, share=None):
        from clap_amd import entities, shard, synth, tiler
        self.rank, self.world, self.use_dist = rank, world, use_dist
        comm_rank = rank
        if block is not None:                                # a test stepping block b of the forest in a smaller world
            rank = block
        self.cam = cam if cam is not None else synth.camera()
        if raw is None:
            raw = synth.entities_chains(chains, depth, seed=2 + rank)
        self.scene = tiler.tiled_scene(raw)[0] if layout == "tiles" else synth.pad_levels(raw)
        self.fr, self.view, _proj = entities.view_calc_frustum(self.cam)
        self.batch = entities.EntityBatch(self.scene, device)
        self.index_base = rank * self.batch.n
        self.pbatch = self.psys = None
        if particles > 0:                                    # whole 1024-particle systems, own RNG stream per rank
            from clap_amd import particles as particles_mod
            n_sys = max(1, particles // 1024)
            self.psys = synth.particle_systems(n_sys=n_sys, count=1024, radius=10.0, velocity=0.005,
                                               dist=synth.PART_DIST_SQRT, seed=40 + rank)
            self.pstate0 = synth.DRAND48_DEFAULT_STATE + rank
            ppos, pvel, pstate = synth.particles_spawn(self.psys, self.pstate0)
            self.pbatch = particles_mod.ParticleBatch(self.psys, ppos, pvel, pstate, device)
        # `share`: a second workload of the same run rides the communicator the first one made (one ncclCommInitRank per run)
        self.xch = shard.VisibleExchange(self.batch, comm_rank, world, device, route=route, share=share) if use_dist else None

    def step(self):
        if self.pbatch is not None:
            self.pbatch.particles_update(self.view)
        if not self.use_dist:
            self.batch.mq_update(self.fr, all_dirty=True)    # one launch for all hierarchy levels (tiles)
            self.batch.compact_visible(self.index_base)      # ordered visible list
            return
        self.xch.begin()
        self.batch.mq_update(self.fr, all_dirty=True)
        self.xch.submit()


def main():
    args = parse()
    if args.c5:
        args.chains, args.depth, args.particles = 250_000, 8, 262_144
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain `python bench.py --gpus N`: be our own launcher
        raise SystemExit(launch_ranks(args))
    if args.dry_launch:
        if "WORLD_SIZE" not in os.environ:
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        return dry_launch()
    import torch
    import torch.distributed as dist
    from clap_amd import _lib, entities, shard, synth, tiler

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    _lib.check(_lib.lib().clapgpu_init(local_rank), "clapgpu_init")
    use_dist = world > 1 or os.environ.get("CLAP_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL on 1 GPU
    wd = InitWatchdog(args.init_timeout, "the collective set-up (process group + ncclCommInitRank)") if use_dist else None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(device))

    cam = synth.camera()
    comps = raw = None
    if args.snapshot:
        from clap_amd import snapshot
        comps = snapshot.load_scene(args.snapshot)
        raw = comps["entities"]
        cam = comps.get("camera", cam)
        args.cpu_frames = 0                                  # the CPU baseline leg times the synthetic workload only
    rs = RankStep(args.chains, args.depth, args.particles, rank, world, device, use_dist=use_dist, route=args.exchange,
                  layout=args.layout, raw=raw, cam=cam)
    scene, batch, pbatch, xch, fr, step = rs.scene, rs.batch, rs.pbatch, rs.xch, rs.fr, rs.step
    n_real, n_pad, index_base = batch.n_real, batch.n, rs.index_base
    proof = None
    if use_dist:                                             # the communicator exists: what does RCCL itself say about the run?
        proof = xch.proof()
        wd.done()
        bad = check_proof(proof, world)
        if bad:                                              # no JSON line for a run that is not what --gpus says
            if rank == 0:
                print(f"[bench] refusing to report: {bad}", file=sys.stderr, flush=True)
            dist.barrier()
            xch.destroy()
            dist.destroy_process_group()
            raise SystemExit(5)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(step_fn):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; the MAX over ranks."""
        for _ in range(args.warmup):
            step_fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        fence()
        mine = worst = time.perf_counter() - t0
        per_rank = None
        if use_dist:
            t = torch.tensor([mine], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst = float(t.item())
            t = torch.tensor([mine], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            per_rank = {"min": float(t.item()) / args.steps * 1e3, "max": worst / args.steps * 1e3}
        return worst, per_rank

    elapsed, rank_ms = timed_leg(step)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_real * args.steps / elapsed
    visible = int((xch.last()[0] if use_dist else batch.visible_count).item())

    # ---- N > 1, no explicit sizing: BASELINE configs[4]'s per-rank share as a SECOND timed leg of the same run, the same
    #      process group and the same communicator (16 M entities + 2 M particles at 8 GPUs).  The headline above stays
    #      configs[1] per GPU -- comparable with the N = 1 line, which is what a scaling curve needs -- and the first run on
    #      an N-GPU node measures the named 8-GPU configuration too, instead of leaving it to a second lease.
    c5 = None
    default_sizing = (args.chains == 125_000 and args.depth == 8 and args.particles == 0 and not args.c5 and not args.snapshot)
    if use_dist and default_sizing and os.environ.get("CLAP_BENCH_NO_C5") != "1":
        rs5 = RankStep(250_000, 8, 262_144, rank, world, device, use_dist=True, route=args.exchange, layout=args.layout, cam=cam,
                       share=xch)
        e5, rank_ms5 = timed_leg(rs5.step)
        vis5 = int(rs5.xch.last()[0].item())
        c5 = {"workload": f"BASELINE configs[4] per-GPU share: {rs5.batch.n_real} entities (250 000 chains x depth 8) + "
                          f"{rs5.pbatch.n_real} particles per GPU = {world * rs5.batch.n_real} entities + "
                          f"{world * rs5.pbatch.n_real} particles on {world} GPU(s); every step: particles_update, entity update + "
                          "cull, allgather of the visibility masks, expansion to the global id list",
              "value": world * rs5.batch.n_real * args.steps / e5, "unit": "entity updates/s",
              "particle_updates_per_s": world * rs5.pbatch.n_real * args.steps / e5,
              "ms_per_step": e5 / args.steps * 1e3, "ms_per_step_per_rank": rank_ms5, "steps": args.steps, "warmup": args.warmup,
              "entities_per_gpu": rs5.batch.n_real, "particles_per_gpu": rs5.pbatch.n_real, "visible": vis5,
              "global_ids": int(rs5.xch.n_pad_all.astype(np.int64).sum()), "scaling": "weak"}
        rs5.xch.destroy()
        rs5 = None
        torch.cuda.empty_cache()

    # ---- roofline pass: HIP events around every launch of the dominant kernel (same stream) ----
    if batch.tiled:
        kernel, launches = "k_entities_tiles<true>", 1
    else:
        kernel, launches = "k_entities_level<true>", batch.n_levels
    # K launches back to back between ONE event pair (events on torch's current stream, the stream the kernels are
    # launched on): an event pair around every single launch adds its own ~5 us of record latency to a 39 us kernel
    # and reads LONGER than the whole step.  The expansion launch is timed the same way, so that
    # mean_launch_us + expand_us ~= ms_per_step * 1000 can be checked from the line itself.
    K = max(args.steps, 20)

    def timed(fn, reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps                       # ms per call

    if batch.tiled:
        lvl_ms = np.asarray([timed(lambda: batch.mq_update(fr, all_dirty=True), K)])
    else:
        lvl_ms = np.asarray([timed(lambda l=l: batch.update_level(l, fr, all_dirty=True), K) for l in range(launches)])
    expand_ms = timed(lambda: batch.compact_visible(index_base), K)
    mean_launch_s = float(lvl_ms.mean()) * 1e-3
    alg_bytes_step = batch.algorithmic_bytes()                                       # 276 B/child, 212 B/root
    alg_bytes_launch = alg_bytes_step / launches
    n_levels = args.depth
    # the committed PMC summary was taken on the default workload: only then is it this launch's traffic
    default_workload = args.chains == 125_000 and args.depth == 8 and not args.snapshot and args.layout == "tiles"

    if rank == 0:
        out = {
            "metric": "entity updates/sec (transform hierarchy + inverse + AABB + frustum cull + visible list)",
            "value": value, "unit": "entity updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"snapshot {os.path.basename(args.snapshot)}: {n_real} entities/GPU, " if args.snapshot
                                    else f"BASELINE configs[4] per-GPU share (--c5; 16 M entities + 2 M particles at 8 GPUs): "
                                         f"{n_real} entities/GPU, {args.chains} chains x depth {args.depth}, " if args.c5
                                    else f"BASELINE configs[1]: {n_real} entities/GPU, {args.chains} chains x depth "
                                         f"{args.depth}, ") + f"{args.layout} SoA layout, all dirty, fused frustum cull + ordered visible "
                                   f"list ({visible} visible" + (" in the gathered global set)" if use_dist else ")"),
                       "entities_per_gpu": n_real, "levels": n_levels,
                       "particles_per_gpu": (pbatch.n_real if pbatch is not None else 0),
                       "exchange": (xch.route
                                    + " of the visibility mask + local expansion to global ids, overlapped with "
                                    "the next frame's update") if use_dist else "none"},
            "roofline": dict(roof(alg_bytes_launch, mean_launch_s, *([kernel] if default_workload else [])),
                             kernel=kernel, launches_per_step=launches,
                             per_launch_us=[float(x) for x in (lvl_ms * 1e3)],
                             launches_timed=K, timing="K launches back to back between one HIP event pair",
                             expand_launch_us=float(expand_ms * 1e3),
                             step_us=float(ms_per_step * 1e3),
                             note="mean_launch_us is launch-to-launch in a dependent stream: the kernel plus the few us between "
                                  "two dependent launches (rocprofv3's kernel-only average is in profiles/); expand_launch_us "
                                  "issued alone is bound by the host's launch rate -- inside a step it costs step_us - "
                                  "mean_launch_us"),
        }
        if proof is not None:                                # the line proves its own N: RCCL's rank count, one PCI bus id per rank
            out["rccl_ranks"] = proof["rccl_ranks"]
            out["devices"] = proof["devices"]
            out["ms_per_step_per_rank"] = rank_ms
        if c5 is not None:
            out["c5"] = c5
        extra = None
        if args.snapshot:
            out["data"] = "snapshot"
            ex = snapshot_characters(comps, raw, device, args.steps, args.warmup)
            if ex:
                extra = {"snapshot_characters": ex}
        if world == 1 and not args.no_extras and not args.snapshot:
            if xch is not None:
                xch.destroy()
            xch = batch = pbatch = step = rs = None           # the extras want the HBM the headline scene holds
            torch.cuda.empty_cache()
            extra = extras(device, testbed=not args.no_testbed)
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(scene, cam, args.cpu_frames)
        # the secondary numbers a reader of a truncated record needs, as a dozen scalars BEFORE the long sections
        out["summary"] = summary(out, extra)
        sec = secondary(out, extra)
        if sec:
            out["roofline"]["secondary"] = sec               # the one dict the driver's record keeps whole
        if extra is not None:
            out["extra"] = extra
        assert c5 is None or list(c5) == C5_KEYS
        out = {k: out[k] for k in LINE_ORDER if k in out} | {k: v for k, v in out.items() if k not in LINE_ORDER}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        if xch is not None:
            xch.destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
