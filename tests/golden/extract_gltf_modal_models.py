#!/usr/bin/env python3
"""Regenerates tests/golden/gltf_modal_models.json from the reference's committed sample glTFs.

The glTFs under /root/reference/glTF_PhysicalAudio/samples embed modal models produced by the reference's real
solver chain (tests/ModalSolveTool.cpp -> GenerateTets + modal::mesh2modes, driven by samples/generate.py:278-335),
so their accessors are golden OUTPUT VECTORS of the path.  This script only decodes data (base64 accessors ->
numbers); it copies no source.  Run in the build container (the GPU box has no /root/reference):
    python tests/golden/extract_gltf_modal_models.py
"""
import base64
import json
import os
import struct
import sys

import numpy as np

ROOT = "/root/reference/glTF_PhysicalAudio/samples"
FILES = ["test/StrikeOne/a_ThreeInstances.gltf", "Pile.gltf", "test/AccelerationNoise/a_SteelBead.gltf"]
FIXTURE = "/root/reference/tests/fixtures/KHR_audio_rigid_bodies.gltf"
NCOMP = {"SCALAR": 1, "VEC3": 3}
FMT = {5126: ("f", 4), 5125: ("I", 4), 5123: ("H", 2), 5121: ("B", 1)}


def accessor(g, buffers, idx):
    a = g["accessors"][idx]
    bv = g["bufferViews"][a["bufferView"]]
    off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    fmt, size = FMT[a["componentType"]]
    n = a["count"] * NCOMP[a["type"]]
    vals = struct.unpack_from("<%d%s" % (n, fmt), buffers[bv["buffer"]], off)
    return list(vals), NCOMP[a["type"]]


def load(path):
    g = json.load(open(path))
    buffers = []
    for b in g.get("buffers", []):
        uri = b.get("uri", "")
        buffers.append(base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else b"")
    return g, buffers


def extract(path, rel):
    g, buffers = load(path)
    ext = g.get("extensions", {}).get("KHR_audio_rigid_bodies", {})
    mats = ext.get("acousticMaterials", [])
    out = []
    for m in ext.get("modalModels", []):
        freqs, _ = accessor(g, buffers, m["frequencies"])
        decay, _ = accessor(g, buffers, m["decayRates"])
        pos, _ = accessor(g, buffers, m["positions"])
        shapes, _ = accessor(g, buffers, m["shapes"])
        idx = accessor(g, buffers, m["indices"])[0] if "indices" in m else []
        npos = len(pos) // 3
        rec = {
            "file": rel, "name": m.get("name", ""), "frequencies": freqs, "decayRates": decay,
            "numPositions": npos, "numTriangles": len(idx) // 3,
            "positionMin": [min(pos[c::3]) for c in range(3)], "positionMax": [max(pos[c::3]) for c in range(3)],
            "massProperties": m.get("massProperties"),
            "material": mats[m["material"]] if "material" in m and m["material"] < len(mats) else None,
            # mode-major shapes; keep mode 0 only (enough to pin layout/scale) to stay small
            "shapeMode0": shapes[: 3 * npos],
        }
        if m.get("name") == "Solved box":
            rec["positions"] = pos
        out.append(rec)
        # the full record -- surface (positions + triangles: the solve tool's INPUT mesh, tests/ModalSolveTool.cpp:84-93) and every
        # mode shape -- goes to the binary companion file
        key = "%s|%s" % (rel, m.get("name", ""))
        FULL[key + "|positions"] = np.array(pos, np.float32).reshape(-1, 3)
        FULL[key + "|indices"] = np.array(idx, np.uint32).reshape(-1, 3)
        FULL[key + "|shapes"] = np.array(shapes, np.float32).reshape(len(freqs), npos, 3)  # mode-major
        FULL[key + "|frequencies"] = np.array(freqs, np.float32)
        FULL[key + "|decayRates"] = np.array(decay, np.float32)
    return out


FULL = {}


def main():
    models = []
    for rel in FILES:
        models += extract(os.path.join(ROOT, rel), rel)
    models += extract(FIXTURE, "tests/fixtures/KHR_audio_rigid_bodies.gltf")
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gltf_modal_models.json")
    def r9(x):  # the accessors are float32: 9 significant digits round-trip them exactly
        if isinstance(x, float):
            return float("%.9g" % x)
        if isinstance(x, list):
            return [r9(v) for v in x]
        if isinstance(x, dict):
            return {k: (v if k in ("massProperties", "material") else r9(v)) for k, v in x.items()}
        return x
    json.dump({"source": "khiner/MeshEditor glTF_PhysicalAudio/samples (decoded accessors)", "models": r9(models)}, open(dst, "w"), separators=(",", ":"))
    print("wrote", dst, len(models), "models", os.path.getsize(dst), "bytes")
    full = os.path.join(os.path.dirname(dst), "gltf_modal_models_full.npz")
    np.savez_compressed(full, **FULL)
    print("wrote", full, os.path.getsize(full), "bytes")
    for m in models:
        print(" ", m["file"], repr(m["name"]), len(m["frequencies"]), "modes", m["numPositions"], "pts", m["frequencies"][:4])


if __name__ == "__main__":
    sys.exit(main())
