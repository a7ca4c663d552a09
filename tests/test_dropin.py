"""The drop-in boundary proven on the reference's OWN objects.

oracle/_ref/clap_dropin (built by oracle/ref/Makefile from the reference's sources where they lie,
plus clap_amd/binding/gpu-scene.c) advances two identical scenes made of the reference's struct mq /
model3dtx / entity3d -- one with the reference's mq_update() + view_entity_in_frustum(), one with
the binding over libclapgpu_scene -> HIP -- through a scripted game (moves, rotations, scales,
visibility and SKIP_CULLING toggles, creations, deletions, re-parenting -- also to parents that come later
in the queue, which the reference reads one frame late -- entities with a foreign update hook that must stay
on the host) and compares mx, inverse_mx, aabb, aabb_center, seq,
parent_seq, xform.updated, the frustum verdict and the camera bounding-volume pick bit for bit
after every frame.  The same binary checks the particle path (`particles` mode): the reference's particles_update
hooks drawing from libc's drand48 against clap_amd/binding/gpu-particles.inc.c.
`anim` mode does the same for skeletal animation (clap_amd/binding/gpu-anim.inc.c), `lights` for the
clustered-lighting masks (clap_amd/binding/gpu-light.inc.c).
The binary needs the reference tree to BUILD (here) and a GPU to RUN.
"""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "clap_dropin")
REF = "/root/reference/core"
# Share of entity updates that must go through the HIP path (the rest runs the reference's own default_update on
# the host on BOTH sides and so proves nothing about the kernels).  `test` is the scripted game, which deliberately
# keeps foreign hooks and children listed before their parents in the mix.
BATCHED_FLOOR = {"test": 0.55}


def test_dropin_checker_is_built_where_the_reference_is():
    if not os.path.isdir(REF):
        pytest.skip("reference tree absent: the checker is prebuilt elsewhere")
    from clap_amd import _lib
    from oracle import refrun
    _lib.build()                                           # the checker links clap_amd/lib/libclapgpu*.so
    refrun.build()
    assert os.access(BIN, os.X_OK)
    # the binding's calls into the product resolve to libclapgpu_scene / libclapgpu, nothing else
    out = subprocess.run(["nm", "-D", "--undefined-only", BIN], capture_output=True, text=True, check=True).stdout
    wanted = {l.split()[-1] for l in out.splitlines() if "clapgpu_" in l}
    assert {"clapgpu_scene_create", "clapgpu_scene_mq_update", "clapgpu_scene_entity_new",
            "clapgpu_scene_entity_transform", "clapgpu_scene_results"} <= wanted
    # the replacement exports the engine's own entry points for the path; the reference's bodies are ref_<name>
    defined = subprocess.run(["nm", "--defined-only", BIN], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1] for l in defined.splitlines() if " T " in l}
    for fn in ("mq_update", "view_entity_in_frustum", "view_calc_frustum", "light_grid_compute", "entity3d_position",
               "entity3d_move", "entity3d_rotate", "entity3d_scale", "entity3d_visible", "entity3d_update", "entity3d_reset",
               "entity3d_delete", "particle_system_position"):
        assert fn in names and "ref_" + fn in names, fn


def _run(*args, env=None):
    if not os.access(BIN, os.X_OK):
        pytest.skip("oracle/_ref/clap_dropin not built (needs the reference tree at build time)")
    p = subprocess.run([BIN, *map(str, args)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, f"clap_dropin {args}: rc {p.returncode}\n{p.stderr[-2000:]}"
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("notify", [False, True], ids=["walk", "notify"])
@pytest.mark.parametrize("n,frames,seed", [(300, 12, 1), (5000, 16, 2), (40000, 8, 3), (2000, 40, 7), (300000, 6, 5)])
def test_binding_matches_reference_mq_update(n, frames, seed, notify):
    """World B is driven through the ENGINE'S OWN NAMES -- mq_update(), view_entity_in_frustum(), entity3d_position /
    _move / _rotate / _scale / _visible -- which gpu-exports.inc.c serves from the binding (link-time substitution: the
    reference's bodies are ref_<name>, world A calls those).  `notify`: the mutators report what they touch and frames
    without creations / deletions / re-parenting run in O(touched + rebuilt) (the 300 000-entity case also takes the
    worker-thread passes)."""
    r = _run("test", n, frames, seed, *(["notify"] if notify else []))
    assert r["mismatches"] == 0
    assert r["notify"] is notify and (r["fast_frames"] > 0) == notify
    assert r["batched_updates"] > 0 and r["host_updates"] > 0      # both halves of the split were exercised
    frac = r["batched_updates"] / (r["batched_updates"] + r["host_updates"])
    assert frac >= BATCHED_FLOOR["test"], f"only {frac:.2f} of the updates went through the device"
    assert r["written_back"] > 0 and r["retiles"] > 0
    assert 0 < r["visible_verdicts_true"]
    assert r["entity3d_update_calls"] > 0          # entity3d_update / entity3d_reset between frames, engine names in world B


@pytest.mark.gpu
@pytest.mark.parametrize("n,frames,seed,steady", [(300, 12, 1, False), (300, 24, 1, True), (5000, 24, 2, True), (40000, 12, 3, True),
                                                  (2000, 40, 7, True), (300000, 6, 5, True)])
def test_drawn_write_back_policy_matches_reference(n, frames, seed, steady):
    """GPU_SCATTER_DRAWN (clap_amd/binding/gpu-scene.h): a fast frame writes back only what is READ -- what the view
    draws, what contains the camera / control position, what has a standing host reader (batched parents of host-class
    children, light carriers, the control entity, entities updated on the spot) -- and leaves the other rebuilt entities
    on the device.  `steady`: three frames out of four the game only moves / turns / scales / hides entities, so most
    frames are fast frames (the scripted game of `test` changes the queue's make-up almost every frame, and a walked
    frame writes everything back).  Checked by clap_dropin: three frames out of four every entity the REFERENCE'S pass
    would draw (ALIVE, VISIBLE, SKIP_CULLING or in the frustum: model.c:959-973) is compared bit for bit WITHOUT any
    fetch -- mx, inverse_mx, aabb, aabb_center, seq, parent_seq, xform.updated, verdict -- and must not be stale; every
    fourth frame and after the last, everything is fetched (gpu_scene_fetch_all) and EVERY entity compared: what was
    left out for several frames catches up exactly, seq counters included (entity3d_update / _reset on stale entities
    and on their parents, re-parenting, deletions and walks in between)."""
    r = _run("test", n, frames, seed, "notify", "drawn", *(["steady"] if steady else []))
    assert r["mismatches"] == 0 and r["scatter"] == "drawn" and r["notify"] is True
    assert r["partial_compare_frames"] > 0 and r["fast_frames"] > 0
    assert r["left_stale"] > 0 and r["stale_seen_by_checker"] > 0, "the policy left nothing out: nothing was tested"
    assert r["fetched_on_view"] > 0, "nothing ever came into view after being left out"
    assert r["entity3d_update_calls"] > 0
    if steady:
        assert r["fast_frames"] >= frames // 2


@pytest.mark.gpu
def test_drawn_policy_without_notifications_is_the_default_policy():
    """A queue that is walked every frame writes everything back whatever the policy says."""
    r = _run("test", 5000, 12, 2, "drawn")
    assert r["mismatches"] == 0 and r["fast_frames"] == 0 and r["left_stale"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_sys,per_sys,frames,seed", [(3, 100, 6, 1), (40, 300, 12, 2), (512, 1024, 8, 3)])
def test_particle_binding_matches_reference_particles_update(n_sys, per_sys, frames, seed):
    """particles_update hooks + libc drand48 (reference) vs gpu_particles_update (binding -> HIP) on the
    reference's own particle_system / particle structs: pos_array, every particle's pos and velocity, the
    billboard matrix and the libc stream position, with emitters moving (attached and free), systems dying
    and appearing, and frames with and without the per-particle scatter-back."""
    r = _run("particles", n_sys, per_sys, frames, seed)
    assert r["mismatches"] == 0 and r["stream_draws_agree"] is True
    assert r["respawns"] > 0 and r["particle_structs_compared"] > 0
    assert r["frames_timed"] == frames - 2 and r["reference_ms_per_frame"] > 0 and r["binding_ms_per_frame_with_particle_structs"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_chars,joints,frames,seed", [(5, 8, 6, 1), (200, 64, 12, 2), (3000, 40, 8, 3), (50, 200, 6, 4)])
def test_animation_binding_matches_reference_animated_update(n_chars, joints, frames, seed):
    """default_update -> animated_update per entity (reference: clock, queue, channels_transform,
    one_joint_transform) vs gpu_mq_update + gpu_anim_update (binding -> HIP) on the reference's own model3d
    (model3d_add_skinning, animation_new / _add_channel), entities from ref_new(entity3d) and queues from
    animation_push_by_name: transforms bit for bit; each joint's T, R, S, its palette matrix and its world position EQUAL
    to the reference's, bit for bit (round 4: the kernel performs the reference's arithmetic); e->animation, the queue
    length, ani_time and the libc drand48 position (the random idle phase of animation_next) exactly; joints outside
    joint 0's tree untouched on both sides.
    Entities riding a character's joint (e->parent_joint, model.c:1626-1641) and their children: those listed after the
    character get the joint transforms of the SAME frame -- on the DEVICE, by the frame's second entity launch behind
    the pose (gpu_scene_run_deferred -> clapgpu_scene_attached_update; counted in attached_batched_updates) --, those
    listed before it the previous frame's, on the host, as in the reference; their matrices, boxes and seq counters
    equal to the reference's."""
    r = _run("anim", n_chars, joints, frames, seed)
    assert r["mismatches"] == 0
    assert r["worst_relative_error"] == 0 and r["differing_objects"] == 0 and r["tolerance"] == 0
    assert r["animation_restarts"] > 0 and r["joint_poses_compared"] == frames * n_chars * joints
    assert r["frames_timed"] == max(frames - 2, 0) and (frames <= 2 or (r["reference_ms_per_frame"] > 0 and r["binding_ms_per_frame"] > 0))
    n_held = n_chars // 3 + 2
    assert r["joint_attached_checks"] == frames * 3 * n_held and r["worst_joint_attached_error"] == 0 and r["joint_attached_differing"] == 0
    # of each triple (rider listed after its character, rider listed before it, plain child of the first) the first and
    # the third go through the device's second launch, every frame (the checker counts them: a rider whose joint number
    # is JOINT_TYPE_MAX is "no joint" to the reference itself and takes the first launch)
    assert r["attached_batched_updates"] == r["attached_expected"] and r["attached_expected"] >= frames * n_held > 0
    assert r["batched_updates"] >= frames * (n_chars + 2 * n_held)


@pytest.mark.gpu
@pytest.mark.parametrize("notify", [False, True], ids=["walk", "notify"])
@pytest.mark.parametrize("n_chars,frames,seed", [(7, 10, 1), (400, 16, 2), (5000, 8, 3)])
def test_character_binding_matches_reference_character_update(n_chars, frames, seed, notify):
    """Body-less characters: the reference's character_update -> default_update per entity against
    the binding (clap_amd/binding/gpu-character.inc.c): the hook's host half -- limbo teleport out of the position
    history, the controlled character's motion reset -- by the reference's OWN character_update with its chained tail
    parked, the transform on the device; props below characters follow.  Entities' matrices, boxes, counters and each
    character's history ring / state / motion fields bit for bit every frame; every update batched."""
    r = _run("characters", n_chars, frames, seed, *(["notify"] if notify else []))
    assert r["mismatches"] == 0
    assert r["batched_updates"] == frames * (n_chars + n_chars // 3 + 1) and r["host_updates"] == 0
    assert (r["fast_frames"] > 0) == notify
    assert r["characters_teleported_by_last_frame"] > 0 or n_chars < 50


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("characters", 5000, 30, 2, "notify"), ("characters", 7, 40, 1, "notify"), ("anim", 500, 64, 30, 5, "notify"),
                                  ("anim", 50, 200, 20, 4, "notify")], ids=["characters-5000", "characters-7", "anim-500x64", "anim-50x200"])
def test_characters_and_animation_under_the_drawn_write_back_policy(args):
    """GPU_SCATTER_DRAWN forced through the environment (GPU_SCENE_SCATTER=drawn, as a maintainer would try it) for the modes
    whose entities have standing host readers: body-less characters (their hook's host half reads the entity), animated
    entities, the props riding their joints through the frame's second launch.  The checker fetches before it compares what
    nobody draws; everything else must be current by the policy's own rules."""
    r = _run(*args, env={"GPU_SCENE_SCATTER": "drawn"})
    assert r["mismatches"] == 0 and r.get("differing_objects", 0) == 0 and r.get("joint_attached_differing", 0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("frames,seed", [(30, 1), (60, 2)])
def test_light_grid_binding_matches_reference_light_grid_compute(frames, seed):
    """light_grid_compute on the reference's own struct light vs gpu_light_grid_compute (binding -> HIP): the
    RGBA32UI masks handed to texture_load, the tile counts after resizes (1080p / 4K / 720p / an odd size at
    8-, 16- and 32-pixel tiles) and the upload call itself."""
    r = _run("lights", frames, seed)
    assert r["mismatches"] == 0 and r["mask_bits_set"] > 0 and r["tiles_compared"] > 0


@pytest.mark.gpu
def test_binding_edge_cases():
    """Hand-made scenes through reference and binding: an empty queue, a single entity, only foreign hooks
    (nothing batched), a parent with 200 children (level-major fallback), a chain 40 deep, entities that were
    never positioned (mx must stay as entity3d_make left it), children of a hooked parent (host), dead
    entities in the list, a skip_aabb model, a few moving neighbours among 1000 (the range-upload path), one moving root of a
    three-level tree, no view to cull against, a queue whose priv is NULL, children listed
    before their parents (the reference's one-frame lag, reproduced on the host) -- also in notification mode with
    moves only, where frames are not walked and a host child listed before its BATCHED parent has to be shown that
    parent's previous-frame matrix and seq although every batched result is already written back --, transforms written
    past the mutators in notification mode with the verification aid on (found, reported, taken in the same frame), and a
    queue emptied and repopulated."""
    r = _run("edge")
    assert r["mismatches"] == 0 and r["cases"] == 23         # (incl. four where a hook deletes another entity while the frame runs)


@pytest.mark.gpu
def test_scene_dumped_by_the_binding_replays_to_the_reference_bits(tmp_path, cuda_device):
    """The loop SURVEY 8f rank 4 is about: a scene made of the reference's objects is dumped by the binding
    (gpu_scene_snapshot_begin -> include/clapgpu_snapshot.h), loaded by clap_amd.snapshot, tiled and run through the
    kernels by the Python harness -- and gives the matrices, AABBs and visible set the REFERENCE computed for it."""
    import ctypes as C

    import numpy as np
    from clap_amd import _lib, entities, snapshot, tiler
    path = str(tmp_path / "scene.clps")
    r = _run("snapshot", 6000, path)
    assert r["rc"] == 0 and r["batched"] == r["entities"] == 6000
    comps = snapshot.load_scene(path)
    raw, exp = comps["entities"], comps["expect"]
    assert (raw["parent"] >= 0).sum() > 1000, "a real hierarchy"
    scene, tl = tiler.tiled_scene(raw)
    fr = _lib.Frustum()
    C.memmove(fr.planes, np.ascontiguousarray(comps["frustum"]["planes"], np.float32).ctypes.data, 96)
    C.memmove(fr.corners, np.ascontiguousarray(comps["frustum"]["corners"], np.float32).ctypes.data, 128)
    batch = entities.EntityBatch(scene, cuda_device)
    batch.mq_update(fr, all_dirty=True)
    batch.compact_visible()
    out = batch.download()
    slot = tl["slot_of"]
    assert np.array_equal(out["mx"][slot].view(np.uint32), exp["mx"].reshape(-1, 16).view(np.uint32))
    assert np.array_equal(out["aabb"][slot].view(np.uint32), exp["aabb"].reshape(-1, 6).view(np.uint32))
    vis = np.zeros(scene["n"], bool)
    vis[out["visible"]] = True
    assert np.array_equal(vis[slot], exp["visible"].astype(bool)) and 0 < vis.sum() < 6000


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("notify",), ("notify", "drawn")], ids=["walk", "notify", "notify-drawn"])
def test_binding_bench_mode_is_consistent(mode):
    """`bench`: both consumers of a frame on both worlds -- one verdict per entity in list order, and _models_render's
    whole per-entity block (model.c:958-992: verdict, LOD pick, the draw's reads of mx / inverse_mx) as the reference's
    loop (world A), as the same loop under the engine's names (world B) and as gpu_scene_select_lod() + the draw list
    per txmodel (world B): the three draw the same entities and read the same bits, every frame, with a drifting camera;
    after the last frame everything is fetched and mx / aabb / seq / parent_seq / cur_lod of every entity compared."""
    r = _run("bench", 20000, 6, 300, *mode)
    assert r["mismatches"] == 0 and r["visible_equal"] is True
    assert r["draw_sets_equal"] is True and r["draw_reads_equal"] is True and r["drawn_per_frame"] > 0
    assert r["binding_mq_update_ms"] > 0 and r["reference_mq_update_ms"] > 0
    assert r["binding_draw_list_ms"] > 0 and r["reference_render_block_ms"] > 0
    if "drawn" in mode:
        assert r["scatter"] == "drawn" and r["left_stale_per_frame"] > 0 and r["fetched_on_view_per_frame"] > 0
    else:
        assert r["left_stale_per_frame"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,frames,permille", [(20000, 30, 100), (100000, 8, 1000), (300000, 5, 100)])
def test_frames_without_notifications_go_by_the_records(n, frames, permille):
    """No notifications at all (the minimal patch): a frame whose queue is the one the last walk met -- checked entity by
    entity through the list nodes, on the workers -- goes by the records instead of chasing the lists on one core, and
    reads what a walk would read; same bits as the serial walk (GPU_SCENE_REPLAY=0), same bits as the reference."""
    r = _run("bench", n, frames, permille)
    assert r["mismatches"] == 0 and r["visible_equal"] is True and r["draw_sets_equal"] is True and r["draw_reads_equal"] is True
    assert r["notify"] is False and r["fast_frames"] == 0 and r["frames_by_the_records"] == frames and r["retiles"] == 0
    w = _run("bench", n, frames, permille, env={"GPU_SCENE_REPLAY": "0"})
    assert w["mismatches"] == 0 and w["frames_by_the_records"] == 0 and w["drawn_per_frame"] == r["drawn_per_frame"]
    # the scripted game without notifications: frames with creations / deletions / re-parenting walk, the others do not
    g = _run("test", min(n, 40000), 24, 7, "steady")
    assert g["mismatches"] == 0 and g["fast_frames"] == 0 and g["frames_by_the_records"] >= 12 and g["retiles"] > 0
    # below 16 384 entities the passes would run on one thread, where the plain walk is the faster frame: walked
    assert _run("bench", 8000, 10, 100)["frames_by_the_records"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [(), ("drawn",)], ids=["all", "drawn"])
@pytest.mark.parametrize("n,frames,churn", [(3000, 30, 5), (20000, 12, 40), (300000, 6, 200)])
def test_entities_made_and_deleted_between_frames_are_not_walked(n, frames, churn, policy):
    """A queue that gains and loses `churn` entities EVERY frame (roots, children of random earlier entities, light
    carriers; leaves deleted): gpu_scene_entity_created / _deleting take them into / out of the standing device layout --
    the frames are not walked and do not re-tile (but for the FIRST such frame, one of the bench's warm-up frames: the queue
    was packed tight until then, and that walk's re-tile makes room for the rest) -- and every consumer still sees the
    reference's bits (verdicts in list order, the render block, the draw list; after the last frame every entity's
    mx / aabb / seq / parent_seq / cur_lod)."""
    r = _run("bench", n, frames, 100, "notify", *policy, "churn", churn)
    assert r["mismatches"] == 0 and r["visible_equal"] is True and r["draw_sets_equal"] is True and r["draw_reads_equal"] is True
    assert r["fast_frames"] == frames and r["retiles"] == 0, r
    assert r["placed_in_layout"] >= 0.97 * frames * churn and r["removed_in_place"] >= 0.97 * frames * churn, r


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [("notify", "drawn", "comeandgo", "plain"), ("notify", "comeandgo", "plain"), ("notify", "drawn", "comeandgo")],
                         ids=["drawn-plain", "all-plain", "drawn-any"])
@pytest.mark.parametrize("n,frames,seed", [(300, 80, 5), (2500, 16, 1), (40000, 12, 3), (200000, 24, 11)])
def test_scripted_game_with_entities_coming_and_going(n, frames, seed, mode):
    """The scripted game with creations and deletions in the frames that are not walked (`comeandgo`), moves, hides,
    host updates and -- every fourth frame -- re-parenting (a walk) in between; `plain`: what the game makes is plain and
    listed behind its parent, i.e. placeable; without it hooked entities and children listed before their parents make most
    creations fall back to a walk.  Every frame compared field by field, seq counters included."""
    r = _run("test", n, frames, seed, *mode)
    assert r["mismatches"] == 0 and r["fast_frames"] > 0
    # (at 40 000 entities the unrestricted game makes ~125 entities a frame, a third of the children listed BEFORE their
    # parents: some creation of every frame needs the walk)
    assert r["removed_in_place"] > 0 and (r["placed_in_layout"] > 0 or ("plain" not in mode and n >= 40000))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [("notify", "drawn", "comeandgo", "plain"), ("notify", "drawn", "steady"), ("notify", "comeandgo"), ("steady",)],
                         ids=["drawn-plain", "drawn-steady", "all", "no-notify"])
def test_worker_passes_forced_from_a_few_hundred_entities(mode):
    """The binding's worker passes (mirror pass, address-list pass, write-back in list order / off the mask, the pass over
    every record of a frame without notifications) start at 12-16 k entities; the thresholds are environment knobs, and with
    them at a few hundred the scripted game -- entities moved twice a frame, updated on the spot, made and deleted -- puts
    every ordering the passes rely on to the test on every run, on a host with many cores (where a soak of this game found a
    child reading its parent's counters half-way through another worker's write-back)."""
    forced = {"GPU_SCENE_MIRROR_PAR_MIN": "300", "GPU_SCENE_SCATTER_PAR_MIN": "200"}
    for n, frames, seed in ((4000, 16, 21), (17000, 10, 22), (60000, 8, 23)):
        r = _run("test", n, frames, seed, *mode, env=forced)
        assert r["mismatches"] == 0, (n, frames, seed, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("launches", [0, 7])
def test_checker_refuses_to_pass_on_the_host_fallback(launches):
    """A device that fails must not look like a slow frame.  `fail <launches>` runs the scripted game with every kernel
    launch after the first <launches> reporting a launch failure (clapgpu_test_fail_after): the engine-side exports
    (gpu-exports.inc.c) then serve mq_update from the reference's host loop -- the two worlds still agree bit for bit,
    which is why the comparison alone would pass -- report it on stderr with clapgpu_last_error() and count it in
    gpu_scene_last_stats()->device_errors; the checker (every mode ends on that counter) exits non-zero."""
    if not os.access(BIN, os.X_OK):
        pytest.skip("oracle/_ref/clap_dropin not built (needs the reference tree at build time)")
    p = subprocess.run([BIN, "fail", str(launches), "2000", "8", "3"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, "the checker passed although the device path failed"
    assert "device_errors" in p.stderr or "did not run" in p.stderr, p.stderr[-1500:]
    assert "gpu_mq_update failed" in p.stderr or "clap gpu binding" in p.stderr or "did not run" in p.stderr, p.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("notify",), ("notify", "drawn", "steady")], ids=["walk", "notify", "notify-drawn-steady"])
@pytest.mark.parametrize("n,frames,seed", [(300, 14, 1), (4000, 12, 2), (12000, 8, 3)])
def test_lod_pick_and_draw_list_match_the_reference_block(n, frames, seed, mode):
    """SURVEY 8f rank 1 through the boundary a CLAP maintainer uses: _models_render's per-entity block (model.c:959-992:
    draw predicate, force_lod, camera-inside-box skip, entity3d_aabb_avg_edge + entity3d_set_lod's clamp) run per pass
    over world A's lists with the reference's own functions, against gpu_scene_select_lod() / gpu_scene_visible() on
    world B (binding -> clapgpu_scene_select_lod -> clapgpu_visible_compact + clapgpu_entities_lod; host-class entities
    by the reference's functions on the host).  e->cur_lod and e->force_lod of EVERY live entity and the set of drawn
    entities agree after every pass -- over a scripted camera path that keeps ending inside entity boxes, second passes
    from another camera (a cull launch for its planes), passes without a camera, LODs forced / released / set through
    entity3d_set_lod (the engine's name in world B), entities hidden, deleted, created and re-parented.  Every entity ON
    the list also holds the reference's mx / inverse_mx / aabb / aabb_center / seq / parent_seq -- under GPU_SCATTER_DRAWN
    too (`drawn`), where a second camera's cull launch has to fetch what its planes bring into view."""
    r = _run("lod", n, frames, seed, *mode)
    assert r["mismatches"] == 0
    assert r["passes"] > frames and r["drawn"] > 0 and r["lod_levels_seen"] >= 3
    assert r["passes_without_a_view"] > 0          # view == NULL: every ALIVE, VISIBLE entity is drawn, whatever the last frustum was
    assert r["forced_or_released"] > 0 and r["drawn_with_camera_inside_box"] > 0
    assert r["batched_updates"] > 0 and r["host_updates"] > 0 and r["notify"] is ("notify" in mode)
    assert r["scatter"] == ("drawn" if "drawn" in mode else "all")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("notify",), ("notify", "drawn", "steady")], ids=["walk", "notify", "notify-drawn-steady"])
@pytest.mark.parametrize("n,frames,seed,passes", [(300, 14, 1, 2), (4000, 12, 2, 4), (30000, 9, 3, 1)])
def test_shadow_passes_with_the_lights_view_before_the_model_pass(n, frames, seed, passes, mode):
    """A frame as pipeline_render renders it: `passes` shadow passes with the LIGHT's view and no camera
    (pipeline-builder.c:34-46, 246-272; model.c:752-760) before the model pass with the camera's.  The light's view is
    registered (gpu_scene_add_view): the update's own launch culls it into a mask of its own, so the shadow passes --
    by one verdict per entity under the engine's name, and by gpu_scene_select_lod's list -- and the model pass behind
    them are all answered without a cull launch; a frame whose light moved after the update costs exactly one.  Draw sets,
    LODs (untouched by a pass without a camera) and the drawn entities' fields against the reference's loop, both
    write-back policies."""
    r = _run("lod", n, frames, seed, *mode, "shadow", passes)
    assert r["mismatches"] == 0
    assert r["shadow_passes"] == frames * passes and r["drawn_by_shadow_passes"] > 0
    assert r["views_culled_by_the_updates"] == 2 * frames                    # the camera's and the light's, every update
    assert r["cull_launches_after_update"] == r["frames_the_light_moved_after_the_update"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [("notify",), ("notify", "drawn")], ids=["notify", "notify-drawn"])
def test_pipeline_shaped_frame_in_bench_mode(mode):
    r = _run("bench", 30000, 6, 100, *mode, "shadow", 2)
    assert r["mismatches"] == 0 and r["shadow_sets_equal"] is True and r["draw_reads_equal"] is True
    assert r["views_culled_per_update"] == 2.0 and r["cull_launches_after_update"] == 0
    assert r["shadow_drawn_per_frame"] > 0 and r["binding_pipeline_frame_draw_list_ms"] > 0


@pytest.mark.gpu
def test_api_orderings_fuzzed_on_the_device():
    """The generated orderings of tests/test_sanitize_host.py::test_api_orderings_fuzzed_under_asan against the real
    library: 200 seeds (the pinned ones among them), four processes at a time."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = [(305, 200), (77, 200), (5016, 1500), (1421, 200)] + [(s, 200) for s in range(1, 181)] + [(s, 1500) for s in range(5001, 5017)]

    def one(job):
        r = _run("fuzz", *job)
        return job[0], r["mismatches"], r["fast_frames"], r["walked_frames"]
    with ThreadPoolExecutor(4) as pool:
        res = list(pool.map(one, jobs))
    assert [s for s, bad, _f, _w in res if bad] == []
    assert sum(f for _s, _b, f, _w in res) > 500 and sum(w for _s, _b, _f, w in res) > 500
