#!/usr/bin/env python3
"""Recovers the tetrahedralisation the REFERENCE used for its committed box-shaped golden models, and writes it to
tests/golden/reference_tets.npz (BUILD CONTAINER ONLY: minutes to an hour of CPU; the tests read the committed file).

Why.  The sample glTFs of the reference hold the real solver chain's outputs (tests/ModalSolveTool.cpp: GenerateTets +
modal::mesh2modes, driven by glTF_PhysicalAudio/samples/generate.py:278-335) for surfaces we can rebuild exactly -- but
the tetrahedraliser is 10 k lines we neither build nor port, and the interior diagonals it picked move the frequencies
of these coarse P2 meshes by ~1e-3: any mesh of ours pins the oracle to the reference at 1e-3, not at the 1e-6 the
parity bar claims.  The bodies "Solved box", "Bar" (12 x 3 x 1 cells) and "Platform" (12 x 1 x 12) are ONE CELL THICK:
every grid point is a surface vertex, the reference added no interior point (its `positions` are exactly the surface
vertices), and a Delaunay-based tetrahedraliser on a grid tiles each cell separately.  So the reference's mesh is one of
finitely many: a diagonal on every interior cell face and, per cell, one of the <= 5 tilings of a cube that fit its six face
diagonals (all 74 tilings of the cube are enumerated below).  The golden carries 10-30 frequencies and every mode shape at
every vertex -- thousands of numbers that respond to each diagonal at the 1e-4 level -- so the mesh is identifiable: a
local search over face flips, with the mismatch between the ORACLE's result on the candidate mesh and the golden as the
objective, walks to a mesh on which the oracle reproduces the golden to float32 round-off (every frequency bit-equal,
shapes to ~1e-9).  No mesh-fitting with ~60 bits of freedom could make a wrong restatement match 3 000 numbers to eight
digits; and the mesh found from the ceramic "Solved box" reproduces the STEEL "Bar" golden (same surface, other density,
stiffness and Poisson ratio -- an independent solve of the reference) bit for bit as well.

    python tests/golden/find_reference_tets.py "Solved box" Platform      # writes / updates reference_tets.npz
"""
import itertools
import os
import pickle
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

# ---- all tetrahedralisations of the cube on its eight corners -----------------------------------------------------------
CORNER = [(x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)]  # local corner id = 4 x + 2 y + z


def _det(a, b, c, d):
    return int(round(np.linalg.det(np.array([np.subtract(b, a), np.subtract(c, a), np.subtract(d, a)]))))


def _side(f, v):
    return 2 if _det(CORNER[f[0]], CORNER[f[1]], CORNER[f[2]], CORNER[v]) > 0 else 1


def _on_hull(f):
    return any(all(CORNER[i][ax] == val for i in f) for ax in range(3) for val in (0, 1))


def _inner_side(g):
    m = np.array([np.subtract(CORNER[g[1]], CORNER[g[0]]), np.subtract(CORNER[g[2]], CORNER[g[0]]), np.subtract((0.5, 0.5, 0.5), CORNER[g[0]])])
    return 2 if np.linalg.det(m) > 0 else 1


def cube_tilings():
    """(faces, tilings): faces[fi] = (corner ids of cube face fi, its two diagonals) for fi = 2 * axis + side; tilings = list of
    (diagonal choice per face, tuple of tets as sorted local corner ids) -- 74 of them, by an advancing front with backtracking:
    a face may carry one tet per side, hull faces only on the inside."""
    faces = []
    for ax in range(3):
        for val in (0, 1):
            idx = [i for i in range(8) if CORNER[i][ax] == val]
            pairs = [(p, q) for p, q in itertools.combinations(idx, 2) if sum(abs(CORNER[p][k] - CORNER[q][k]) for k in range(3)) == 2]
            faces.append((idx, pairs))
    out = []

    def advance(used, tiling, choice):
        open_faces = [(f, b) for f, b in used.items() if b in (1, 2)]
        if not open_faces:
            out.append((choice, tuple(sorted(tiling))))
            return
        f, b = min(open_faces)
        for v in range(8):
            if v in f or _det(CORNER[f[0]], CORNER[f[1]], CORNER[f[2]], CORNER[v]) == 0 or _side(f, v) != 3 ^ b:
                continue
            t = tuple(sorted(f + (v,)))
            marks = []
            for i in range(4):
                g = tuple(x for k, x in enumerate(t) if k != i)
                bit = _side(g, t[i])
                if _on_hull(g) and (g not in used or bit != _inner_side(g)):
                    break
                if used.get(g, 0) & bit:
                    break
                marks.append((g, bit))
            else:
                saved = {g: used.get(g) for g, _ in marks}
                for g, bit in marks:
                    used[g] = used.get(g, 0) | bit
                advance(used, tiling + [t], choice)
                for g, _ in marks:
                    if saved[g] is None:
                        del used[g]
                    else:
                        used[g] = saved[g]

    for choice in itertools.product((0, 1), repeat=6):
        used = {}
        for (idx, pairs), ch in zip(faces, choice):
            p, q = pairs[ch]
            for o in idx:
                if o not in (p, q):
                    g = tuple(sorted((p, q, o)))
                    used[g] = 3 ^ _inner_side(g)  # the outside of a hull triangle is taken
        advance(used, [], choice)
    return faces, out


FACES, TILINGS = cube_tilings()
assert len(TILINGS) == 74 and sorted({len(t) for _, t in TILINGS}) == [5, 6]


def _is_main(fi, ch):
    """Does diagonal `ch` of cube face `fi` pass through the face's corner with the smallest coordinates?"""
    ax = fi // 2
    p, q = FACES[fi][1][ch]
    others = [k for k in range(3) if k != ax]
    return all(CORNER[p][k] == 0 for k in others) or all(CORNER[q][k] == 0 for k in others)


# pattern of the six face diagonals (0 = through the face's smallest corner, 1 = the other one) -> the tilings that carry it
BY_PATTERN = {}
for _choice, _t in TILINGS:
    BY_PATTERN.setdefault(tuple(0 if _is_main(fi, ch) else 1 for fi, ch in enumerate(_choice)), []).append(_t)


# ---- a one-cell-thick grid body and the meshes it admits ---------------------------------------------------------------
class Grid:
    def __init__(self, positions32, triangles):
        self.pos32 = positions32
        axes = [np.unique(positions32[:, k]) for k in range(3)]
        self.n = [len(a) - 1 for a in axes]
        ijk = np.stack([np.searchsorted(axes[k], positions32[:, k]) for k in range(3)], 1)
        self.vid = {tuple(x): i for i, x in enumerate(ijk)}
        self.cubes = [(i, j, k) for i in range(self.n[0]) for j in range(self.n[1]) for k in range(self.n[2])]
        for cube in self.cubes:
            for c in CORNER:
                assert (cube[0] + c[0], cube[1] + c[1], cube[2] + c[2]) in self.vid, "a grid point is not a surface vertex: the body is not one cell thick"
        edges = set()
        for t in triangles:
            for e in range(3):
                a, b = int(t[e]), int(t[(e + 1) % 3])
                edges.add((min(a, b), max(a, b)))
        self.fixed, self.free = {}, []  # hull faces: the diagonal of the input surface; inner faces: unknowns
        for cube in self.cubes:
            for fi in range(6):
                key = self.face_key(cube, fi)
                if key in self.fixed or key in self.free:
                    continue
                ax, plane = fi // 2, cube[fi // 2] + fi % 2
                if plane in (0, self.n[ax]):
                    bit = None
                    for ch in (0, 1):
                        p, q = FACES[fi][1][ch]
                        gp, gq = self.corner(cube, p), self.corner(cube, q)
                        if (min(gp, gq), max(gp, gq)) in edges:
                            bit = 0 if _is_main(fi, ch) else 1
                    assert bit is not None
                    self.fixed[key] = bit
                else:
                    self.free.append(key)
        self.cube_faces = {cube: [self.face_key(cube, fi) for fi in range(6)] for cube in self.cubes}
        self.face_cubes = {}
        for cube in self.cubes:
            for key in self.cube_faces[cube]:
                self.face_cubes.setdefault(key, []).append(cube)

    def corner(self, cube, local):
        c = CORNER[local]
        return self.vid[(cube[0] + c[0], cube[1] + c[1], cube[2] + c[2])]

    @staticmethod
    def face_key(cube, fi):
        ax = fi // 2
        o = [k for k in range(3) if k != ax]
        return (ax, cube[ax] + fi % 2, cube[o[0]], cube[o[1]])

    def pattern(self, cube, bits):
        return tuple(self.fixed[k] if k in self.fixed else bits[k] for k in self.cube_faces[cube])

    def tets(self, bits, sel):
        out = []
        for cube in self.cubes:
            options = BY_PATTERN.get(self.pattern(cube, bits))
            if not options:
                return None
            for t in options[sel.get(cube, 0) % len(options)]:
                out.append([self.corner(cube, l) for l in t])
        tets = np.array(out, np.uint32)
        p = self.pos32.astype(np.float64)
        vol = np.einsum("ij,ij->i", np.cross(p[tets[:, 1]] - p[tets[:, 0]], p[tets[:, 2]] - p[tets[:, 0]]), p[tets[:, 3]] - p[tets[:, 0]])
        flip = vol < 0
        tets[flip, 0], tets[flip, 1] = tets[flip, 1].copy(), tets[flip, 0].copy()
        return tets


class Mismatch:
    """Oracle result on a candidate mesh against the golden: sum of squared relative frequency errors + 1e-2 x squared relative
    shape errors (modes of nearly equal frequency compared as subspaces)."""

    def __init__(self, grid, material, gold_freqs, gold_shapes, max_freq):
        from oracle import pyoracle as po
        self.po, self.grid, self.mat = po, grid, material
        po.set_threads(int(os.environ.get("ORACLE_THREADS", "1")))  # these systems are tiny: a thread team only adds start-up cost
        self.gf, self.gs = gold_freqs.astype(np.float64), gold_shapes.astype(np.float64)
        # the solve tool's configuration (tests/ModalSolveTool.cpp:66-71; generate.py asks for 30 modes)
        self.cfg = po.default_config(num_modes=30, num_fem_modes=45, max_mode_freq=max_freq)
        self.evals = 0

    def solve(self, tets):
        self.evals += 1
        return self.po.mesh2modes(self.grid.pos32.astype(np.float64), tets, self.po.material(*self.mat), self.grid.pos32, config=self.cfg)

    def __call__(self, tets, detail=False):
        r = self.solve(tets)
        k = len(self.gf)
        if len(r.freqs) < k:
            return (1e9, None, None) if detail else 1e9
        df = (r.freqs[:k] - self.gf) / self.gf
        sh = np.transpose(r.shapes[:, :k, :], (1, 0, 2)).astype(np.float64)
        es = np.zeros(k)
        m = 0
        while m < k:
            e = m + 1
            while e < k and (self.gf[e] - self.gf[e - 1]) < 3e-3 * self.gf[e]:
                e += 1
            a, b = sh[m:e].reshape(e - m, -1).T, self.gs[m:e].reshape(e - m, -1).T
            x, *_ = np.linalg.lstsq(a, b, rcond=None)
            es[m:e] = np.linalg.norm(a @ x - b, axis=0) / np.linalg.norm(b, axis=0)
            m = e
        j = float((df ** 2).sum() + 1e-2 * (es ** 2).sum())
        return (j, df, es) if detail else j


def search(grid, mismatch, bits, sel, rng, sweeps, log=print, checkpoint=lambda bits, sel: None):
    best = mismatch(grid.tets(bits, sel))
    log("start: mismatch %.3e" % best)
    for sweep in range(sweeps):
        improved = False
        order = list(grid.free)
        rng.shuffle(order)
        for key in order:  # one inner face at a time, with every fitting tiling of the two cells it touches
            cubes = grid.face_cubes[key]
            trial = dict(bits)
            trial[key] = 1 - bits[key]
            options = [BY_PATTERN.get(grid.pattern(c, trial)) for c in cubes]
            if any(o is None for o in options):
                continue
            found = None
            for combo in itertools.product(*[range(len(o)) for o in options]):
                tsel = dict(sel)
                tsel.update(zip(cubes, combo))
                j = mismatch(grid.tets(trial, tsel))
                if j < best:
                    best, found = j, tsel
            if found is not None:
                bits, sel, improved = trial, found, True
        if best > 1e-12:
            cubes = list(grid.cubes)
            rng.shuffle(cubes)
            for cube in cubes:  # two or more faces of one cell at once (a single flip may leave a cell without a tiling)
                free = [k for k in grid.cube_faces[cube] if k in bits]
                found = None
                for flips in itertools.product((0, 1), repeat=len(free)):
                    if sum(flips) < 2:
                        continue
                    trial = dict(bits)
                    for k, f in zip(free, flips):
                        if f:
                            trial[k] = 1 - bits[k]
                    touched = sorted({c for k, f in zip(free, flips) if f for c in grid.face_cubes[k]})
                    options = [BY_PATTERN.get(grid.pattern(c, trial)) for c in touched]
                    if any(o is None for o in options):
                        continue
                    combos = itertools.product(*[range(len(o)) for o in options]) if np.prod([len(o) for o in options]) <= 12 else [tuple(0 for _ in options)]
                    for combo in combos:
                        tsel = dict(sel)
                        tsel.update(zip(touched, combo))
                        j = mismatch(grid.tets(trial, tsel))
                        if j < best:
                            best, found = j, (trial, tsel)
                if found is not None:
                    bits, sel = found
                    improved = True
                    checkpoint(bits, sel)
        for cube in grid.cubes:  # the tiling of one cell among those that fit its faces
            options = BY_PATTERN[grid.pattern(cube, bits)]
            for s in range(len(options)):
                if len(options) > 1 and s != sel.get(cube, 0) % len(options):
                    tsel = dict(sel)
                    tsel[cube] = s
                    j = mismatch(grid.tets(bits, tsel))
                    if j < best:
                        best, sel, improved = j, tsel, True
        log("sweep %d: mismatch %.3e after %d solves" % (sweep, best, mismatch.evals))
        checkpoint(bits, sel)
        if not improved or best < 1e-14:
            break
    return bits, sel, best


# golden model -> (key in gltf_modal_models_full.npz, material (density, Young, Poisson, alpha, beta), MaxModeFreq, mesh name)
MODELS = {
    "Solved box": ("test/StrikeOne/a_ThreeInstances.gltf|Solved box", (2700.0, 7.2e10, 0.19, 6.0, 1e-7), 16000.0, "box_12x3x1"),
    "Bar": ("Pile.gltf|Bar", (7850.0, 2.0e11, 0.29, 5.0, 3e-8), 16000.0, "box_12x3x1"),
    "Platform": ("Pile.gltf|Platform", (2700.0, 7.2e10, 0.19, 6.0, 1e-7), 16000.0, "platform_12x1x12"),
}


def main(names):
    full = np.load(os.path.join(HERE, "gltf_modal_models_full.npz"))
    dst = os.path.join(HERE, "reference_tets.npz")
    for name in names:
        key, material, max_freq, mesh_name = MODELS[name]
        grid = Grid(full[key + "|positions"], full[key + "|indices"])
        mismatch = Mismatch(grid, material, full[key + "|frequencies"], full[key + "|shapes"], max_freq)
        state = os.path.join("/tmp", "find_reference_tets_%s.pkl" % mesh_name)
        bits, sel = pickle.load(open(state, "rb")) if os.path.exists(state) else ({k: 0 for k in grid.free}, {})
        t0 = time.time()
        bits, sel, best = search(grid, mismatch, bits, sel, np.random.default_rng(0), sweeps=int(os.environ.get("SWEEPS", "12")),
                                 checkpoint=lambda b, s: pickle.dump((b, s), open(state, "wb")))
        j, df, es = mismatch(grid.tets(bits, sel), detail=True)
        print("%s: cells %s, %d inner faces, mismatch %.3e in %.0f s; max |df/f| %.2e, max shape error %.2e" % (name, grid.n, len(grid.free), j, time.time() - t0, np.abs(df).max(), es.max()))
        if np.abs(df).max() < 2e-7:
            found = dict(np.load(dst)) if os.path.exists(dst) else {}  # (read now: another run may have added a mesh meanwhile)
            found[mesh_name] = grid.tets(bits, sel)
            np.savez_compressed(dst, **found)
            print("  -> %s[%s]" % (dst, mesh_name))
        else:
            print("  not identified (frequencies differ beyond float32 round-off): nothing written")


if __name__ == "__main__":
    main(sys.argv[1:] or ["Solved box"])
