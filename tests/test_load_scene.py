"""scene.json + glTF loader (include/clapgpu_load.h; scene.c:1318-1884, gltf.c:666-1331 restated in C).

CPU: tests/golden/scene_fixture/ (composed by tests/golden/make_scene_fixture.py from the file formats) is loaded
and every extracted array compared with what the files were composed from; the loader's rules (skipped entries,
defaults, mesh choice, dropped channels / animations, light slots) are checked one by one.  GPU: the loaded
scene is replayed through the kernels -- entity update, pose, skinning -- against the oracle.
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from clap_amd import _lib, snapshot, synth
from oracle import binding as ob
from helpers import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "scene_fixture")
LIBDIR = os.path.join(ROOT, "clap_amd", "lib")

E_VISIBLE, E_CHAR, E_PHYS, E_BODY, E_LIGHT, E_ARM, E_ANIM, E_SKIPCULL, E_DIRTY, E_ATTACHED, E_ALIVE = (
    1, 2, 1 << 4, 1 << 5, 1 << 8, 1 << 12, 1 << 13, 1 << 14, 1 << 16, 1 << 17, 1 << 31)


@pytest.fixture(scope="module")
def loaded(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("load") / "fixture.clps")
    snapshot.load_scene_json(os.path.join(FIX, "scene.json"), p)
    return snapshot.load_scene(p), np.load(os.path.join(FIX, "expected.npz")), json.load(open(os.path.join(FIX, "scene.json")))


def test_header_functions_all_bound_and_exported():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "clapgpu_load.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(clapgpu_\w+)\s*\(", text)))
    assert sorted(snapshot.LOAD_SYMBOLS) == declared
    out = subprocess.run(["nm", "-D", "--defined-only", snapshot.SCENE_LIB_PATH], capture_output=True, text=True, check=True)
    assert set(declared) <= {l.split()[-1] for l in out.stdout.splitlines() if " T " in l}


def test_skinned_model_extraction(loaded):
    comps, exp, _js = loaded
    m = comps["model0"]
    assert m["nr_joints"] == 12 and m["n_verts"] == 96
    for k in ("position", "normal", "joints", "weights", "invmx"):
        assert np.array_equal(m[k], exp[f"hero_{k}"]) and m[k].dtype == exp[f"hero_{k}"].dtype, k
    assert np.array_equal(m["joint_parent"], exp["hero_parent"])
    # bind = mat4x4_invert(invmx) with the engine's arithmetic: the oracle's restatement of linmath agrees bit for bit
    assert np.array_equal(m["bind"].view(np.uint32), ob.skeleton_bind(dict(nr_joints=12, invmx=m["invmx"])).view(np.uint32))
    # root pose: mat4x4_from_quat(rotation of the node named like the skin), then its translation
    q, t = exp["hero_root_q"], exp["hero_root_t"]
    rp = m["root_pose"].reshape(4, 4)
    assert np.array_equal(rp[3], np.asarray([t[0], t[1], t[2], 1.0], np.float32))
    x, y, z, w = (float(v) for v in q)
    assert np.allclose(rp[:3, :3], np.asarray([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)],
                                                [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)],
                                                [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]]), atol=1e-6)
    # "armature": roles by joint NAME; a name that matches nothing leaves the role unset
    names = list(exp["hero_joint_names"])
    jt = m["joint_types"]
    assert jt[1] == names.index("bone.07") and jt[4] == names.index("bone.10") and jt[5] == names.index("bone.11")
    assert jt[0] == -1 and jt[2] == -1 and jt[3] == -1
    # animations: the channel on a non-joint node is dropped, the animation without joint channels is deleted
    assert m["n_anims"] == 2
    for a in range(2):
        for k in ("ch_target", "ch_path", "ch_nr", "ch_time_off", "ch_data_off", "times", "data"):
            assert np.array_equal(m[f"a{a}_{k}"], exp[f"hero_a{a}_{k}"]), (a, k)
        assert m[f"a{a}_time_end"][0] == exp[f"hero_a{a}_time_end"]
    assert "a2_times" not in m


def test_static_models_and_mesh_choice(loaded):
    comps, exp, _js = loaded
    # crate: fix_origin moves the origin to the bottom centre of the AABB before the box is taken
    pos = exp["crate_position"]
    lo, hi = pos.min(0), pos.max(0)
    c = np.asarray([(lo[0] + hi[0]) / np.float32(2), lo[1], (lo[2] + hi[2]) / np.float32(2)], np.float32)
    moved = pos - c
    assert np.array_equal(comps["model1"]["position"], moved)
    assert np.array_equal(comps["entities"]["model_aabb"][1], np.concatenate([moved.min(0), moved.max(0)]))
    # lamp: two meshes, the scene's root node names mesh 1
    assert np.array_equal(comps["model2"]["position"], exp["lamp_position"])
    assert np.array_equal(comps["entities"]["model_aabb"][2], np.concatenate([exp["lamp_position"].min(0), exp["lamp_position"].max(0)]))
    # hero: mesh 0 of two (the other is "collision"), root node = the first listed that is not "Light"
    assert np.array_equal(comps["entities"]["model_aabb"][0],
                          np.concatenate([exp["hero_position"].min(0), exp["hero_position"].max(0)]))
    assert not comps["entities"]["model_skip"].any()


def test_entities_follow_the_scene_file(loaded):
    comps, _exp, js = loaded
    e = comps["entities"]
    rows = [(mi, ent, "character" in m) for mi, m in enumerate(js["model"]) for ent in m.get("character", m.get("entity", []))]
    assert e["n"] == len(rows) == 51
    L = snapshot.lib()
    L.clapgpu_quat_from_angles.argtypes = [C_FP, C_INT, C_FP]
    L.clapgpu_quat_from_angles.restype = None
    names = {}
    for i, (mi, ent, is_char) in enumerate(rows):
        if "name" in ent and ent["name"] not in names:
            names[ent["name"]] = i
        assert e["model"][i] == mi
        want_flags = E_ALIVE | E_VISIBLE | E_DIRTY
        if mi == 0:
            want_flags |= E_ARM | E_ANIM
        if is_char:
            want_flags |= E_CHAR | E_SKIPCULL
        ps, rot = np.asarray([0, 0, 0, 1], np.float32), np.asarray([0, 0, 0, 1], np.float32)
        parent, pj, done = -1, -1, False
        if "attach" in ent:
            if ent["attach"] not in names:
                done = True                                  # mq_find_entity fails: the rest of the entry is skipped
            else:
                parent = names[ent["attach"]]
        if not done and "attach_joint" in ent:
            role = {"head": 1, "foot_left": 2, "foot_right": 3, "hand_left": 4, "hand_right": 5}[ent["attach_joint"]]
            j = comps["model0"]["joint_types"][role]
            if j < 0:
                done = True
            else:
                pj = int(j)
        if not done and "rotate" in ent:
            q = np.zeros(4, np.float32)
            a = np.asarray(ent["rotate"], np.float32)
            L.clapgpu_quat_from_angles(a.ctypes.data_as(C_FP), 1, q.ctypes.data_as(C_FP))
            rot = q
        p = ent.get("position")
        if not done and p is not None and len(p) >= 3:
            ps[:3] = np.asarray(p[:3], np.float32)
            if len(p) >= 4:
                ps[3] = np.float32(p[3])
                if len(p) >= 5:
                    q = np.zeros(4, np.float32)
                    a = np.asarray([0.0, np.float32(float(np.float32(p[4])) * np.pi / 180.0), 0.0], np.float32)
                    L.clapgpu_quat_from_angles(a.ctypes.data_as(C_FP), 0, q.ctypes.data_as(C_FP))
                    rot = q
                if "light_color" in ent:
                    want_flags |= E_LIGHT
                phys = js["model"][mi].get("physics")
                if phys and phys.get("geom", "sphere") != "sphere":
                    want_flags |= E_PHYS | (E_BODY if phys.get("type", "body") == "body" else 0)
                if parent >= 0 and pj >= 0:
                    want_flags |= E_ATTACHED
        assert e["parent"][i] == parent and e["parent_joint"][i] == pj, (i, ent)
        assert np.array_equal(e["pos_scale"][i], ps), (i, ent, e["pos_scale"][i], ps)
        assert np.array_equal(e["rot"][i].view(np.uint32), rot.view(np.uint32)), (i, ent)
        assert e["flags"][i] == want_flags, (i, ent, hex(int(e["flags"][i])), hex(want_flags))
    assert not e["seqs"].any()
    # the side tables
    assert list(comps["characters"]["entity"]) == [0, 1] and list(comps["characters"]["speed"]) == [1.5, 1.5]
    assert list(comps["characters"]["can_jump"]) == [1, 1] and list(comps["characters"]["can_dash"]) == [0, 0]
    at = comps["attach"]
    assert list(at["entity"]) == [names["torch"], names["hat"]] and list(at["parent"]) == [names["player"], names["npc"]]
    b = comps["bodies"]
    assert list(b["entity"][:2]) == [0, 1] and list(b["geom_class"][:2]) == [1, 1] and list(b["phys_type"][:2]) == [0, 0]
    assert b["mass"][0] == 70.0 and b["radius"][0] == 0.4 and b["length"][0] == 1.1 and b["yoffset"][0] == 0.95
    assert b["bounce"][0] == 0.1 and np.isinf(b["bounce_vel"][0])
    assert len(b["entity"]) == 2 + 40 and set(b["geom_class"][2:]) == {2} and set(b["phys_type"][2:]) == {1}
    assert b["bounce_vel"][2] == 0.2 and b["mass"][2] == 1.0          # defaults where the file is silent


def test_lights_take_slots_in_file_order(loaded):
    comps, _exp, js = loaded
    li, car = comps["lights"], comps["carriers"]
    e = comps["entities"]
    assert li["nr_lights"] == 5 and list(li["active"][:6]) == [1, 1, 1, 1, 1, 0]
    lamp = {ent.get("name"): ent for ent in js["model"][2]["entity"]}
    assert list(car["light"]) == [0, 1, 2]
    torch, spot, street = (int(v) for v in car["entity"])
    # torch: point light (attenuation given), position = LOCAL entity position + offset (light_update_from_entity)
    assert np.array_equal(li["pos"][0], e["pos_scale"][torch][:3] + np.asarray(lamp["torch"]["light_offset"], np.float32))
    assert np.array_equal(car["offset"][0], np.asarray(lamp["torch"]["light_offset"], np.float32))
    assert np.array_equal(li["color"][0], np.asarray([4, 3, 1], np.float32)) and li["is_dir"][0] == 0
    assert np.array_equal(li["attenuation"][0], np.asarray([1.0, 0.35, 0.44], np.float32))
    # spot: cutoff in radians, directional flag, direction = -(rotation * +Z)
    assert li["is_dir"][1] == 1 and li["cutoff"][1] == np.float32(30.0 * np.pi / 180.0)
    assert np.allclose(li["dir"][1], [np.sqrt(0.5), 0, -np.sqrt(0.5)], atol=1e-6)
    assert np.array_equal(li["pos"][1], e["pos_scale"][spot][:3]) and np.array_equal(li["attenuation"][1], [1, 0, 0])
    assert li["is_dir"][2] == 0 and np.array_equal(li["pos"][2], e["pos_scale"][street][:3])
    # scene-level lights come after the entities' (file order): a directional one, then a point light
    assert li["is_dir"][3] == 1 and np.array_equal(li["pos"][3], [100, 200, 50]) and np.array_equal(li["dir"][3], np.asarray([0.4, 0.8, 0.2], np.float32))
    assert li["is_dir"][4] == 0 and np.array_equal(li["attenuation"][4], np.asarray([1.0, 0.7, 1.8], np.float32))
    assert np.array_equal(li["ambient"], np.asarray([0.1, 0.1, 0.15], np.float32))
    assert np.array_equal(li["shadow_tint"], np.asarray([0.2, 0.1, 0.3], np.float32))


def test_errors_are_reported_not_crashed_on(tmp_path):
    bad = tmp_path / "bad"
    bad.mkdir()
    cases = {
        "not json": "{",
        "model is not an array": '{"model": 3}',
        "model without gltf": '{"model": [{"name": "x"}]}',
        "missing asset": '{"model": [{"name": "x", "gltf": "nope.glb"}]}',
        "light without color": '{"light": [{"position": [1, 2, 3]}]}',
    }
    for what, text in cases.items():
        p = bad / "scene.json"
        p.write_text(text)
        with pytest.raises(_lib.ClapGpuError) as ei:
            snapshot.load_scene_json(str(p), str(bad / "out.clps"))
        assert str(ei.value), what
        assert not (bad / "out.clps").exists(), what
    # a truncated GLB, a GLB whose JSON lacks required members, a skin whose joint nodes are numbered past the table
    glb = open(os.path.join(FIX, "hero.glb"), "rb").read()
    (bad / "scene.json").write_text('{"model": [{"name": "h", "gltf": "h.glb", "entity": [{"position": [0, 0, 0, 1]}]}]}')
    for cut in (10, 40, len(glb) // 2, len(glb) - 4):
        (bad / "h.glb").write_bytes(glb[:cut])
        with pytest.raises(_lib.ClapGpuError):
            snapshot.load_scene_json(str(bad / "scene.json"), str(bad / "out.clps"))
    rng = np.random.default_rng(5)
    for _ in range(200):                                        # bit flips anywhere: refused or loaded, never a crash
        b = bytearray(glb)
        for k in rng.integers(0, len(b), 3):
            b[int(k)] ^= 1 << int(rng.integers(0, 8))
        (bad / "h.glb").write_bytes(bytes(b))
        try:
            snapshot.load_scene_json(str(bad / "scene.json"), str(bad / "out.clps"))
        except _lib.ClapGpuError:
            pass
    # indices and offsets that are negative, fractional, non-finite or huge (a cast of those is undefined behaviour and
    # (unsigned)-1 would index wildly): such bufferViews / accessors are skipped, the asset is refused or loads, no crash
    import struct
    jl = struct.unpack_from("<I", glb, 12)[0]
    doc = json.loads(glb[20:20 + jl].decode())
    pos_acc = doc["meshes"][0]["primitives"][0]["attributes"]["POSITION"]
    for field, where, value in (("bufferView", "accessors", -1), ("bufferView", "accessors", 1e300), ("byteOffset", "accessors", -5),
                                ("count", "accessors", -3), ("count", "accessors", 2.5), ("buffer", "bufferViews", -1),
                                ("byteOffset", "bufferViews", -1e9), ("byteOffset", "bufferViews", 1.8e19),
                                ("byteLength", "bufferViews", -1)):
        d = json.loads(json.dumps(doc))
        idx = pos_acc if where == "accessors" else d["accessors"][pos_acc]["bufferView"]
        d[where][idx][field] = value
        js = json.dumps(d, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        bin_chunk = glb[20 + jl:]
        out = struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + len(bin_chunk)) + struct.pack("<II", len(js), 0x4E4F534A) + js + bin_chunk
        (bad / "h.glb").write_bytes(out)
        try:
            snapshot.load_scene_json(str(bad / "scene.json"), str(bad / "out.clps"))
        except _lib.ClapGpuError:
            pass


def test_loader_under_sanitizers(tmp_path):
    """The loader + snapshot writer built with AddressSanitizer + UBSan (host code), run over the fixture and over
    damaged copies of its files."""
    exe = str(tmp_path / "test_load_c")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-DTEST_LOAD_NO_GPU", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "test_load.c"),
                    os.path.join(ROOT, "clap_amd", "host", "clapgpu_load.c"), os.path.join(ROOT, "clap_amd", "host", "clapgpu_snapshot.c"),
                    "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, FIX, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout


import ctypes as _C
C_FP, C_INT = _C.POINTER(_C.c_float), _C.c_int


# ------------------------------------------------------------------------------------------- GPU replay
@pytest.mark.gpu
def test_loaded_scene_replays_on_the_gpu(loaded, cuda_device):
    """The loaded scene through the kernels: entity update (tile layout) bit-exact against the oracle, then the
    characters' pose (both animations, several times) and skinning, bit for bit."""
    from clap_amd import animation, entities, tiler
    comps, _exp, _js = loaded
    raw = dict(comps["entities"])
    raw["flags"] = raw["flags"] & ~np.uint32(E_ATTACHED)          # joint attachments need the palette first: second pass below
    raw["model_lod"] = np.zeros((3, 2), np.uint8)
    scene, tl = tiler.tiled_scene(raw)
    batch = entities.EntityBatch(scene, cuda_device)
    fr, _v, _p = entities.view_calc_frustum(synth.camera(pos=(0, 5, 40)))
    batch.mq_update(fr, all_dirty=True)
    out = batch.download()
    st = ob.entity_state(scene)
    ob.entities_update(scene, st)
    for k in ("mx", "inv_mx", "aabb", "center"):
        assert np.array_equal(out[k].view(np.uint32), st[k].view(np.uint32)), k

    (sk, anims, mesh), = snapshot.skinned_models(comps).values()
    chars = comps["characters"]["entity"]
    n, J, V = len(chars), sk["nr_joints"], mesh["n_verts"]
    slots = tl["slot_of"][chars]
    char_mx = out["mx"][slots]
    model = animation.SkinnedModel(sk, anims, mesh=mesh, bind=sk["bind"], device=cuda_device)
    cb = animation.CharacterBatch(model, n, np.zeros((J, 10), np.float32), char_mx, vert_first=np.zeros(n, np.uint32),
                                  vert_count=np.full(n, V, np.uint32))
    trs = np.zeros((n, J, 10), np.float32)
    reach = sk["order"]
    for a_id, times in ((0, [0.0, 0.4]), (1, [1.3, 2.5]), (0, [1.25, 0.1])):
        for t0 in times:
            t = np.asarray([t0, t0 * 0.5 + 0.05], np.float32)[:n]
            cb.anim.fill_(a_id)
            cb.set_frame_times(t)
            cb.pose_update()
            cb.skin()
            got = cb.download()
            jt, _g, jp = ob.pose(sk, anims[a_id], t, char_mx, trs)
            assert bits_equal(got["joint_transforms"][:, reach], jt[:, reach])              # bit-exact (round 4)
            assert bits_equal(got["joint_pos"][:, reach], jp[:, reach])
            op, on = ob.skin(mesh, np.zeros(n, np.uint32), np.full(n, V, np.uint32), jt)
            assert bits_equal(got["out_position"], op) and bits_equal(got["out_normal"], on)


@pytest.mark.gpu
def test_c_program_loads_and_replays_the_fixture(tmp_path, cuda_device):
    """tests/c/test_load.c: load scene.json + hero.glb from C, replay entities through the C host mirror (bit-exact
    against the oracle) and the characters' pose / skinning through the flat ABI (equal values)."""
    ob.lib()                                                  # builds oracle/_build/libclap_oracle.so
    exe = str(tmp_path / "test_load")
    odir = os.path.join(ROOT, "oracle", "_build")
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
                    os.path.join(ROOT, "tests", "c", "test_load.c"), "-o", exe, "-L", LIBDIR, "-lclapgpu_scene", "-lclapgpu",
                    "-L", odir, "-lclap_oracle", "-lm", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{odir}"], check=True)
    r = subprocess.run([exe, FIX, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout


def _glb_patch(path_in, path_out, patch):
    """Rewrite bytes of a GLB's BIN chunk: patch(doc, view_of) -> [(byte offset in BIN, bytes)]."""
    import struct
    raw = bytearray(open(path_in, "rb").read())
    jl = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + jl].decode())
    bin0 = 20 + jl + 8

    def span(acc_index):
        a = doc["accessors"][acc_index]
        v = doc["bufferViews"][a["bufferView"]]
        return v.get("byteOffset", 0) + a.get("byteOffset", 0), a["count"]
    for off, data in patch(doc, span):
        raw[bin0 + off:bin0 + off + len(data)] = data
    open(path_out, "wb").write(bytes(raw))
    return doc


def test_loader_flags_unnormalised_weights_and_non_strict_key_times(tmp_path):
    """weight_sum_max_dev: the shader's total_local_pos.w is the weight sum (model.vert:36-38; clapgpu_skin_batch.out_w).
    a<k>_ch_nonstrict / key_times_nonstrict: equal or descending neighbouring key times make channel_time_to_idx's
    bracket depend on its cursor (model.c:1266-1288, 1310), which the stateless device search does not have."""
    src = os.path.join(FIX, "hero.glb")
    clean = str(tmp_path / "clean.clps")
    snapshot.load_gltf(src, clean)
    m = snapshot.load_scene(clean)["model0"]
    assert m["key_times_nonstrict"] == 0 and not m["a0_ch_nonstrict"].any() and not m["a1_ch_nonstrict"].any()
    assert 0.0 <= float(m["weight_sum_max_dev"][0]) <= 2e-7             # the fixture's Dirichlet weights, rounded to fp32

    picked = {}

    def patch(doc, span):
        out = []
        # vertex 5: weights scaled by 1.5; vertex 9: all zero
        w_acc = doc["meshes"][0]["primitives"][0]["attributes"]["WEIGHTS_0"]
        off, _n = span(w_acc)
        w = np.frombuffer(open(src, "rb").read(), np.uint8)             # re-read: offsets are relative to BIN
        import struct
        jl = struct.unpack_from("<I", w, 12)[0]
        bin0 = 20 + jl + 8
        w5 = np.frombuffer(w[bin0 + off + 5 * 16:bin0 + off + 6 * 16].tobytes(), np.float32) * np.float32(1.5)
        out.append((off + 5 * 16, w5.astype(np.float32).tobytes()))
        out.append((off + 9 * 16, np.zeros(4, np.float32).tobytes()))
        picked["sum5"] = float(w5.astype(np.float32).sum(dtype=np.float32))
        # the first joint channel of animation 0 with >= 3 keys: key 2 := key 1 (equal neighbours);
        # the first of animation 1 with >= 3 keys: key 1 > key 2 (descending)
        for ai, mode in ((0, "equal"), (1, "descending")):
            an = doc["animations"][ai]
            for ci, ch in enumerate(an["channels"]):
                if ch["target"]["node"] >= 12:
                    continue
                t_off, cnt = span(an["samplers"][ch["sampler"]]["input"])
                if cnt >= 3:
                    t = np.frombuffer(w[bin0 + t_off:bin0 + t_off + 4 * cnt].tobytes(), np.float32).copy()
                    if mode == "equal":
                        t[2] = t[1]
                    else:
                        t[1], t[2] = t[2] + np.float32(0.01), t[1]
                    out.append((t_off, t.tobytes()))
                    picked[ai] = sum(1 for c in an["channels"][:ci] if c["target"]["node"] < 12)   # index among kept channels
                    break
        return out

    bad = str(tmp_path / "bad.glb")
    _glb_patch(src, bad, patch)
    snap = str(tmp_path / "bad.clps")
    snapshot.load_gltf(bad, snap)                                       # kept and flagged, not rejected
    m2 = snapshot.load_scene(snap)["model0"]
    dev = float(m2["weight_sum_max_dev"][0])
    assert abs(dev - max(abs(picked["sum5"] - 1.0), 1.0)) <= 1e-6       # the all-zero vertex: |0 - 1| = 1
    assert m2["key_times_nonstrict"] == 2
    assert m2["a0_ch_nonstrict"][picked[0]] == 1 and m2["a0_ch_nonstrict"].sum() == 1
    assert m2["a1_ch_nonstrict"][picked[1]] == 1 and m2["a1_ch_nonstrict"].sum() == 1
