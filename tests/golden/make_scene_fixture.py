#!/usr/bin/env python3
"""Writes tests/golden/scene_fixture/: a hand-made CLAP scene in the engine's own file formats.

Nothing of the reference is executed or copied: the files are composed from the formats the engine reads --
scene.json as scene.c:1318-1884 walks it ("model" [{name, gltf, physics, armature, entity / character [...]}],
"light" [...]) and glTF 2.0 assets as gltf.c:666-1331 reads them (a GLB with a skin, two animations and a
collision mesh; a plain .gltf with its buffer in a base64 data URI).  `expected.npz` keeps the source arrays
the files were composed from, so that tests can check what the loader extracts against what was put in.

    python tests/golden/make_scene_fixture.py
"""
import base64
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "scene_fixture")
F32 = np.float32


class Buf:
    """One glTF binary buffer with bufferViews / accessors appended as data is added."""

    def __init__(self):
        self.data = bytearray()
        self.views, self.accessors = [], []

    def add(self, arr, type_, comptype):
        raw = np.ascontiguousarray(arr).tobytes()
        while len(self.data) % 4:
            self.data.append(0)
        self.views.append({"buffer": 0, "byteOffset": len(self.data), "byteLength": len(raw)})
        self.data += raw
        self.accessors.append({"bufferView": len(self.views) - 1, "componentType": comptype, "count": int(arr.shape[0]),
                               "type": type_})
        return len(self.accessors) - 1


def rigid(rng, n, spread):
    """n column-major rigid mat4 (rotation + translation) as float32 [n][16]."""
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    m = np.zeros((n, 4, 4))
    m[:, 0, 0] = 1 - 2 * (y * y + z * z); m[:, 0, 1] = 2 * (x * y + z * w); m[:, 0, 2] = 2 * (x * z - y * w)
    m[:, 1, 0] = 2 * (x * y - z * w); m[:, 1, 1] = 1 - 2 * (x * x + z * z); m[:, 1, 2] = 2 * (y * z + x * w)
    m[:, 2, 0] = 2 * (x * z + y * w); m[:, 2, 1] = 2 * (y * z - x * w); m[:, 2, 2] = 1 - 2 * (x * x + y * y)
    m[:, 3, :3] = rng.uniform(-spread, spread, (n, 3))
    m[:, 3, 3] = 1
    return m.reshape(n, 16).astype(F32)


def hero_glb(rng):
    """Skinned character: 12 joints (nodes 0..11, so node number == the engine's node->joint table index),
    an armature node named like the skin (root pose), a mesh node, a 'collision' mesh, two animations of which
    one also drives a non-joint node (that channel must be dropped), and a third that drives no joint at all."""
    J, V = 12, 96
    parent = np.array([-1, 0, 0, 1, 1, 2, 3, 3, 5, 5, 8, 9], np.int32)
    joint_order = np.array([0, 2, 1, 3, 5, 4, 6, 7, 9, 8, 10, 11], np.int32)      # skin.joints: joint j is node joint_order[j]
    node_of_joint = joint_order
    joint_of_node = np.argsort(joint_order).astype(np.int32)
    b = Buf()
    pos = rng.uniform(-1, 1, (V, 3)).astype(F32)
    pos[:, 1] = rng.uniform(0, 2, V)
    nor = rng.normal(size=(V, 3)); nor = (nor / np.linalg.norm(nor, axis=1, keepdims=True)).astype(F32)
    jnt = rng.integers(0, J, (V, 4)).astype(np.uint8)
    wgt = rng.dirichlet(np.ones(4), V).astype(F32)
    idx = rng.integers(0, V, 3 * 40).astype(np.uint16)
    a_pos, a_nor = b.add(pos, "VEC3", 5126), b.add(nor, "VEC3", 5126)
    a_jnt, a_wgt = b.add(jnt, "VEC4", 5121), b.add(wgt, "VEC4", 5126)
    a_idx = b.add(idx, "SCALAR", 5123)
    cpos = rng.uniform(-0.5, 0.5, (8, 3)).astype(F32)
    cidx = np.arange(6, dtype=np.uint16)
    a_cpos, a_cidx = b.add(cpos, "VEC3", 5126), b.add(cidx, "SCALAR", 5123)
    invmx = rigid(rng, J, 1.0)
    a_inv = b.add(invmx, "MAT4", 5126)

    nodes = []
    for n in range(J):                                   # node n is joint joint_of_node[n]
        j = int(joint_of_node[n])
        kids = [int(node_of_joint[c]) for c in range(J) if parent[c] == j]
        nd = {"name": f"bone.{n:02d}", "translation": [float(v) for v in rng.uniform(-0.3, 0.3, 3).astype(F32)],
              "rotation": [0.0, 0.0, 0.0, 1.0], "scale": [1.0, 1.0, 1.0]}
        if kids:
            nd["children"] = kids
        nodes.append(nd)
    root_q = np.array([0.0, 0.38268343, 0.0, 0.92387953], F32)
    root_t = np.array([0.25, 0.0, -0.5], F32)
    nodes.append({"name": "Armature", "rotation": [float(v) for v in root_q], "translation": [float(v) for v in root_t],
                  "children": [int(node_of_joint[0])]})                         # node 12
    nodes.append({"name": "HeroMesh", "mesh": 0, "skin": 0})                    # node 13
    nodes.append({"name": "Prop", "translation": [1.0, 2.0, 3.0]})              # node 14: animated but no joint
    nodes.append({"name": "Light"})                                             # node 15: never the root

    anims, exp_anims = [], []
    for ai, (name, keys, t_end) in enumerate((("Walk", 5, 1.25), ("Idle", 3, 2.0))):
        samplers, channels = [], []
        e = dict(target=[], path=[], nr=[], toff=[], doff=[], times=[], data=[])
        t_at = d_at = 0
        targets = [(n, p) for n in range(J) for p in range(3) if rng.uniform() < 0.7]
        if ai == 0:
            targets.insert(3, (14, 0))                   # a channel on the non-joint node: dropped by the loader
            # (no second channel for one (joint, path): the engine keeps ONE keyframe cursor per joint and path,
            #  model.c:1305-1311, so what two such channels give depends on the cursor's history -- not a parity case)
        for n, p in targets:
            k = int(rng.integers(2, keys + 1))
            t = np.sort(rng.uniform(0, t_end, k)).astype(F32)
            t[0] = 0.0
            if (n, p) == targets[0]:
                t[-1] = t_end
            t = np.unique(t)
            k = t.shape[0]
            if p == 1:
                q = rng.normal(size=(k, 4)); d = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F32)
            elif p == 0:
                d = rng.uniform(-0.5, 0.5, (k, 3)).astype(F32)
            else:
                d = rng.uniform(0.8, 1.25, (k, 3)).astype(F32)
            a_t = b.add(t, "SCALAR", 5126)
            a_d = b.add(d, "VEC4" if p == 1 else "VEC3", 5126)
            samplers.append({"input": a_t, "output": a_d, "interpolation": "LINEAR"})
            channels.append({"sampler": len(samplers) - 1,
                             "target": {"node": n, "path": ("translation", "rotation", "scale")[p]}})
            if n < J:
                e["target"].append(int(joint_of_node[n])); e["path"].append(p); e["nr"].append(k)
                e["toff"].append(t_at); e["doff"].append(d_at); e["times"].append(t); e["data"].append(d.ravel())
                t_at += k; d_at += d.size
        anims.append({"name": name, "samplers": samplers, "channels": channels})
        exp_anims.append(dict(ch_target=np.asarray(e["target"], np.uint32), ch_path=np.asarray(e["path"], np.uint32),
                              ch_nr=np.asarray(e["nr"], np.uint32), ch_time_off=np.asarray(e["toff"], np.uint32),
                              ch_data_off=np.asarray(e["doff"], np.uint32), times=np.concatenate(e["times"]),
                              data=np.concatenate(e["data"]),
                              time_end=F32(max(float(t[-1]) for t in e["times"]))))
    # an animation that touches no joint (an exported curve): the engine deletes it again
    a_t = b.add(np.asarray([0.0, 1.0], F32), "SCALAR", 5126)
    a_d = b.add(np.zeros((2, 3), F32), "VEC3", 5126)
    anims.append({"name": "Curve", "samplers": [{"input": a_t, "output": a_d, "interpolation": "LINEAR"}],
                  "channels": [{"sampler": 0, "target": {"node": 14, "path": "translation"}}]})

    doc = {
        "asset": {"version": "2.0", "generator": "tests/golden/make_scene_fixture.py"},
        "scene": 0,
        "scenes": [{"name": "Scene", "nodes": [15, 13, 12]}],
        "nodes": nodes,
        "materials": [{"name": "skin", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.6, 0.5, 1.0], "metallicFactor": 0.1,
                                                                 "roughnessFactor": 0.7}}],
        "meshes": [{"name": "Hero", "primitives": [{"attributes": {"POSITION": a_pos, "NORMAL": a_nor, "JOINTS_0": a_jnt,
                                                                   "WEIGHTS_0": a_wgt}, "indices": a_idx, "material": 0}]},
                   {"name": "collision", "primitives": [{"attributes": {"POSITION": a_cpos}, "indices": a_cidx, "material": 0}]}],
        "skins": [{"name": "Armature", "inverseBindMatrices": a_inv, "joints": [int(v) for v in joint_order]}],
        "animations": anims,
        "accessors": b.accessors, "bufferViews": b.views, "buffers": [{"byteLength": len(b.data)}],
    }
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    bin_ = bytes(b.data) + b"\0" * (-len(b.data) % 4)
    total = 12 + 8 + len(js) + 8 + len(bin_)
    glb = struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js \
        + struct.pack("<II", len(bin_), 0x004E4942) + bin_
    exp = dict(parent=parent, invmx=invmx, position=pos, normal=nor, joints=jnt, weights=wgt, root_q=root_q, root_t=root_t,
               joint_names=[f"bone.{int(node_of_joint[j]):02d}" for j in range(J)])
    return glb, exp, exp_anims


def static_gltf(rng, name, n_verts, lo, hi, two_meshes=False):
    """A static mesh in a plain .gltf with its buffer as a base64 data URI."""
    b = Buf()
    pos = rng.uniform(lo, hi, (n_verts, 3)).astype(F32)
    idx = rng.integers(0, n_verts, 3 * 8).astype(np.uint16)
    a_pos, a_idx = b.add(pos, "VEC3", 5126), b.add(idx, "SCALAR", 5123)
    meshes = [{"name": name, "primitives": [{"attributes": {"POSITION": a_pos}, "indices": a_idx, "material": 0}]}]
    nodes = [{"name": name, "mesh": 0}]
    if two_meshes:                                       # the root node names mesh 1: that is the one instantiated
        pos2 = rng.uniform(2 * lo, 2 * hi, (n_verts // 2, 3)).astype(F32)
        a_pos2 = b.add(pos2, "VEC3", 5126)
        meshes.append({"name": name + ".hull", "primitives": [{"attributes": {"POSITION": a_pos2}, "indices": a_idx, "material": 0}]})
        nodes = [{"name": name + ".root", "mesh": 1}, {"name": name, "mesh": 0}]
        pos = pos2
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"name": "Scene", "nodes": [0]}], "nodes": nodes,
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1]}}], "meshes": meshes,
           "accessors": b.accessors, "bufferViews": b.views,
           "buffers": [{"byteLength": len(b.data),
                        "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(b.data)).decode()}]}
    return json.dumps(doc, indent=1).encode(), pos


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260)
    glb, hero, hero_anims = hero_glb(rng)
    open(os.path.join(OUT, "hero.glb"), "wb").write(glb)
    crate_js, crate_pos = static_gltf(rng, "Crate", 24, -0.5, 0.5)
    open(os.path.join(OUT, "crate.gltf"), "wb").write(crate_js)
    lamp_js, lamp_pos = static_gltf(rng, "Lamp", 16, -0.2, 0.9, two_meshes=True)
    open(os.path.join(OUT, "lamp.gltf"), "wb").write(lamp_js)

    crates = []
    for k in range(40):
        p = rng.uniform(-30, 30, 3).astype(F32)
        ent = {"position": [float(p[0]), float(abs(p[1]) / 10), float(p[2]), float(F32(rng.uniform(0.5, 2.0)))]}
        if k % 3 == 0:
            ent["position"].append(float(F32(rng.uniform(-400, 400))))          # y rotation in degrees, beyond +-180 too
        if k % 5 == 0:
            ent["rotate"] = [float(F32(v)) for v in rng.uniform(-200, 200, 3)]
        if k == 7:
            ent["name"] = "marked crate"
        crates.append(ent)
    crates.append({"name": "no position: stays at the origin"})
    crates.append({"position": [1.0, 2.0]})                                      # too short: position untouched
    scene = {
        "name": "fixture",
        "model": [
            {"name": "hero", "gltf": "hero.glb", "speed": 1.5, "can_jump": True,
             "armature": {"head": "bone.07", "hand_left": "bone.10", "hand_right": "bone.11", "foot_left": "no such bone"},
             "physics": {"geom": "capsule", "type": "body", "mass": 70.0, "radius": 0.4, "length": 1.1, "yoffset": 0.95,
                         "bounce": 0.1},
             "animations": {"motion": "Walk", "idle": "Idle"},
             "character": [{"name": "player", "position": [0.0, 0.0, 5.0, 1.0, 90.0]},
                           {"name": "npc", "position": [12.0, 0.5, -3.0, 1.25], "rotate": [0.0, 45.0, 10.0]}]},
            {"name": "crate", "gltf": "crate.gltf", "fix_origin": True,
             "physics": {"geom": "trimesh", "type": "geom", "bounce_vel": 0.2},
             "entity": crates},
            {"name": "lamp", "gltf": "lamp.gltf", "physics": {"mass": 2.0},      # class defaults to sphere: the engine creates no body
             "entity": [{"name": "torch", "attach": "player", "attach_joint": "hand_right", "position": [0.05, 0.1, 0.0, 0.5],
                         "rotate": [10.0, 20.0, 30.0], "light_color": [4.0, 3.0, 1.0], "light_offset": [0.0, 0.4, 0.0],
                         "light_attenuation": [1.0, 0.35, 0.44]},
                        {"name": "hat", "attach": "npc", "attach_joint": "head", "position": [0.0, 0.2, 0.0, 1.0]},
                        {"name": "orphan", "attach": "nobody", "position": [9.0, 9.0, 9.0, 1.0]},
                        {"name": "sock", "attach": "npc", "attach_joint": "foot_left", "position": [1.0, 1.0, 1.0, 1.0]},
                        {"name": "bag", "attach": "player", "position": [0.0, 1.0, -0.3, 0.8]},
                        {"name": "spot", "position": [5.0, 6.0, 7.0, 1.0, -45.0], "light_color": [1.0, 1.0, 1.0],
                         "light_cutoff": 30.0},
                        {"name": "street lamp", "position": [-8.0, 0.0, 4.0, 2.0], "light_color": [2.0, 2.0, 1.5],
                         "light_attenuation": [1.0, 0.09, 0.032]}]},
        ],
        "light": [
            {"ambient_color": [0.1, 0.1, 0.15]},
            {"position": [100.0, 200.0, 50.0], "color": [1.0, 0.95, 0.9], "direction": [-0.4, -0.8, -0.2]},
            {"shadow_tint": [0.2, 0.1, 0.3]},
            {"position": [3.0, 4.0, 5.0], "color": [0.5, 0.5, 2.0], "attenuation": [1.0, 0.7, 1.8]},
        ],
    }
    open(os.path.join(OUT, "scene.json"), "w").write(json.dumps(scene, indent=2) + "\n")
    arrays = {f"hero_{k}": v for k, v in hero.items() if k != "joint_names"}
    arrays["hero_joint_names"] = np.asarray(hero["joint_names"])
    for a, an in enumerate(hero_anims):
        for k, v in an.items():
            arrays[f"hero_a{a}_{k}"] = v
    arrays["crate_position"], arrays["lamp_position"] = crate_pos, lamp_pos
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **arrays)
    print("wrote", OUT, {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
