// ORACLE (test infrastructure): see sparse_chol.h.
#include "sparse_chol.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>

namespace oracle {
namespace {
constexpr int LeafSize = 96; // unknowns below which a subdomain is factorised as one dense front
constexpr int PanelWidth = 48;

// F(i,j) -= sum_k P(i,k) * P(j,k) over the trailing lower triangle i >= j >= c0, where P is the
// kb-wide panel starting at column k0 of the same column-major array.  Register-blocked: 4 columns x 8 rows.
void trailing_update(double *__restrict F, int ld, int f, int k0, int kb, int c0) {
    constexpr int JB = 4, IB = 8;
    const int nblocks = (f - c0) / JB;
    // column blocks are independent (each writes its own four columns): deal them to the OpenMP team; the arithmetic
    // per entry is the same with any thread count
#pragma omp parallel for schedule(dynamic, 2) if (nblocks >= 64)
    for (int jb = 0; jb < nblocks; ++jb) {
        const int j = c0 + jb * JB;
        int i = j;
        for (; i + IB <= f; i += IB) {
            double a0[IB] = {}, a1[IB] = {}, a2[IB] = {}, a3[IB] = {};
            for (int k = 0; k < kb; ++k) {
                const double *pk = F + size_t(k0 + k) * ld;
                const double *pi = pk + i;
                const double b0 = pk[j], b1 = pk[j + 1], b2 = pk[j + 2], b3 = pk[j + 3];
                for (int ii = 0; ii < IB; ++ii) {
                    const double v = pi[ii];
                    a0[ii] += v * b0;
                    a1[ii] += v * b1;
                    a2[ii] += v * b2;
                    a3[ii] += v * b3;
                }
            }
            double *c0p = F + size_t(j) * ld + i, *c1p = c0p + ld, *c2p = c1p + ld, *c3p = c2p + ld;
            for (int ii = 0; ii < IB; ++ii) {
                c0p[ii] -= a0[ii];
                c1p[ii] -= a1[ii];
                c2p[ii] -= a2[ii];
                c3p[ii] -= a3[ii];
            }
        }
        for (; i < f; ++i) {
            for (int jj = 0; jj < JB; ++jj) {
                double s = 0;
                for (int k = 0; k < kb; ++k) s += F[size_t(k0 + k) * ld + i] * F[size_t(k0 + k) * ld + j + jj];
                F[size_t(j + jj) * ld + i] -= s;
            }
        }
    }
    for (int j = c0 + nblocks * JB; j < f; ++j) {
        for (int i = j; i < f; ++i) {
            double s = 0;
            for (int k = 0; k < kb; ++k) s += F[size_t(k0 + k) * ld + i] * F[size_t(k0 + k) * ld + j];
            F[size_t(j) * ld + i] -= s;
        }
    }
}

// Cholesky of the leading nj columns of the f x f column-major lower-stored F, leaving the Schur
// complement in the trailing (f-nj) x (f-nj) block.
bool partial_cholesky(double *F, int f, int nj) {
    const int ld = f;
    for (int k0 = 0; k0 < nj; k0 += PanelWidth) {
        const int kb = std::min(PanelWidth, nj - k0);
        for (int j = k0; j < k0 + kb; ++j) {
            double *cj = F + size_t(j) * ld;
            for (int k = k0; k < j; ++k) {
                const double *ck = F + size_t(k) * ld;
                const double ljk = ck[j];
                if (ljk == 0) continue;
                for (int i = j; i < f; ++i) cj[i] -= ck[i] * ljk;
            }
            if (!(cj[j] > 0)) return false;
            const double d = std::sqrt(cj[j]);
            cj[j] = d;
            const double inv = 1.0 / d;
            for (int i = j + 1; i < f; ++i) cj[i] *= inv;
        }
        if (k0 + kb < f) trailing_update(F, ld, f, k0, kb, k0 + kb);
    }
    return true;
}
} // namespace

bool MultifrontalCholesky::factorize(const CscLower &a) {
    n = a.n;
    fronts.clear();
    L.clear();
    Flops = 0;
    // Full symmetric structure with values (both triangles), columns in the original numbering.
    std::vector<int64_t> ptr(n + 1, 0);
    for (int c = 0; c < n; ++c) {
        for (int64_t p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
            const int r = a.row[p];
            ++ptr[c + 1];
            if (r != c) ++ptr[r + 1];
        }
    }
    for (int c = 0; c < n; ++c) ptr[c + 1] += ptr[c];
    std::vector<int> idx(ptr[n]);
    std::vector<double> val(ptr[n]);
    {
        std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
        for (int c = 0; c < n; ++c) {
            for (int64_t p = a.colptr[c]; p < a.colptr[c + 1]; ++p) {
                const int r = a.row[p];
                idx[cur[c]] = r;
                val[cur[c]++] = a.val[p];
                if (r != c) {
                    idx[cur[r]] = c;
                    val[cur[r]++] = a.val[p];
                }
            }
        }
    }

    // ---- Nested dissection by breadth-first level structures ----
    struct TNode {
        std::vector<int> sep;
        int child[2]{-1, -1};
    };
    std::vector<TNode> tree;
    std::vector<int> label(n, 0), level(n, 0);
    int next_label = 1;
    std::vector<int> queue;
    queue.reserve(n);

    // Breadth-first search restricted to vertices carrying `lab`; returns the visiting order with level numbers.
    auto bfs = [&](int start, int lab, int visited_lab) {
        queue.clear();
        queue.push_back(start);
        label[start] = visited_lab;
        level[start] = 0;
        for (size_t head = 0; head < queue.size(); ++head) {
            const int v = queue[head];
            for (int64_t p = ptr[v]; p < ptr[v + 1]; ++p) {
                const int u = idx[p];
                if (label[u] != lab) continue;
                label[u] = visited_lab;
                level[u] = level[v] + 1;
                queue.push_back(u);
            }
        }
    };

    std::function<int(std::vector<int> &)> dissect = [&](std::vector<int> &verts) -> int {
        const int node = int(tree.size());
        tree.emplace_back();
        if (int(verts.size()) <= LeafSize) {
            tree[node].sep = std::move(verts);
            return node;
        }
        const int lab = next_label++;
        for (const int v : verts) label[v] = lab;
        // Pseudo-peripheral start: two sweeps, each restarting from the last vertex reached.
        int start = verts[0];
        int cur = lab;
        for (int sweep = 0; sweep < 2; ++sweep) {
            const int vis = next_label++;
            bfs(start, cur, vis);
            start = queue.back();
            // Restore the label of unreached vertices' set id: reached ones now carry `vis`; relabel them back.
            for (const int v : queue) label[v] = lab;
        }
        const int vis = next_label++;
        bfs(start, lab, vis);
        std::vector<int> part_a, part_b, sep;
        if (queue.size() < verts.size()) {
            // Disconnected: the reached component against the rest, no separator.
            part_a = queue;
            for (const int v : verts)
                if (label[v] == lab) part_b.push_back(v);
        } else {
            const int nlevels = level[queue.back()] + 1;
            if (nlevels < 3) {
                tree[node].sep = std::move(verts);
                return node;
            }
            // The level where the running count crosses half, kept strictly inside the structure.
            std::vector<int> count(nlevels, 0);
            for (const int v : queue) ++count[level[v]];
            int m = 0;
            size_t run = 0;
            for (; m < nlevels; ++m) {
                run += count[m];
                if (run * 2 >= verts.size()) break;
            }
            m = std::clamp(m, 1, nlevels - 2);
            for (const int v : queue) {
                if (level[v] < m) {
                    part_a.push_back(v);
                } else if (level[v] > m) {
                    part_b.push_back(v);
                } else {
                    bool touches_b = false;
                    for (int64_t p = ptr[v]; p < ptr[v + 1] && !touches_b; ++p) {
                        const int u = idx[p];
                        touches_b = label[u] == vis && level[u] == m + 1;
                    }
                    (touches_b ? sep : part_a).push_back(v);
                }
            }
        }
        verts.clear();
        verts.shrink_to_fit();
        tree[node].sep = std::move(sep);
        const int ca = dissect(part_a);
        const int cb = dissect(part_b);
        tree[node].child[0] = ca;
        tree[node].child[1] = cb;
        return node;
    };
    std::vector<int> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    const int root = n > 0 ? dissect(all) : -1;

    // ---- Postorder numbering; one front per non-empty separator / leaf ----
    perm.assign(n, 0);
    iperm.assign(n, 0);
    int next = 0;
    std::function<std::vector<int>(int)> number = [&](int node) -> std::vector<int> {
        std::vector<int> tops;
        for (const int c : tree[node].child) {
            if (c < 0) continue;
            auto t = number(c);
            tops.insert(tops.end(), t.begin(), t.end());
        }
        auto &sep = tree[node].sep;
        if (sep.empty()) return tops;
        std::sort(sep.begin(), sep.end());
        Front fr;
        fr.j0 = next;
        fr.nj = int(sep.size());
        for (const int v : sep) {
            perm[next] = v;
            iperm[v] = next++;
        }
        fr.children = std::move(tops);
        fronts.push_back(std::move(fr));
        return {int(fronts.size()) - 1};
    };
    if (root >= 0) number(root);
    tree.clear();

    // ---- Symbolic: update-row structure of each front ----
    std::vector<int> mark(n, -1);
    size_t total = 0;
    for (int t = 0; t < int(fronts.size()); ++t) {
        auto &fr = fronts[t];
        const int jend = fr.j0 + fr.nj;
        for (int j = fr.j0; j < jend; ++j) {
            const int c = perm[j];
            for (int64_t p = ptr[c]; p < ptr[c + 1]; ++p) {
                const int i = iperm[idx[p]];
                if (i >= jend && mark[i] != t) {
                    mark[i] = t;
                    fr.urows.push_back(i);
                }
            }
        }
        for (const int c : fr.children) {
            for (const int i : fronts[c].urows) {
                if (i >= jend && mark[i] != t) {
                    mark[i] = t;
                    fr.urows.push_back(i);
                }
            }
        }
        std::sort(fr.urows.begin(), fr.urows.end());
        fr.loff = total;
        total += size_t(fr.nj + fr.urows.size()) * fr.nj;
    }
    L.assign(total, 0.0);

    // ---- Numeric multifrontal factorisation ----
    std::vector<std::unique_ptr<std::vector<double>>> updates(fronts.size());
    std::vector<int> loc(n, -1);
    std::vector<double> F;
    for (int t = 0; t < int(fronts.size()); ++t) {
        auto &fr = fronts[t];
        const int nj = fr.nj, nu = int(fr.urows.size()), f = nj + nu;
        F.assign(size_t(f) * f, 0.0);
        for (int k = 0; k < nj; ++k) loc[fr.j0 + k] = k;
        for (int k = 0; k < nu; ++k) loc[fr.urows[k]] = nj + k;
        for (int k = 0; k < nj; ++k) {
            const int j = fr.j0 + k, c = perm[j];
            double *col = F.data() + size_t(k) * f;
            for (int64_t p = ptr[c]; p < ptr[c + 1]; ++p) {
                const int i = iperm[idx[p]];
                if (i >= j) col[loc[i]] += val[p];
            }
        }
        for (const int c : fr.children) {
            const auto &cu = fronts[c].urows;
            const int cn = int(cu.size());
            const double *U = updates[c]->data();
            for (int b = 0; b < cn; ++b) {
                double *col = F.data() + size_t(loc[cu[b]]) * f;
                const double *ucol = U + size_t(b) * cn;
                for (int r = b; r < cn; ++r) col[loc[cu[r]]] += ucol[r];
            }
            updates[c].reset();
        }
        if (!partial_cholesky(F.data(), f, nj)) return false;
        Flops += double(nj) * f * f; // order-of-magnitude count (exact: sum over pivots of (f-k)^2)
        double *panel = L.data() + fr.loff;
        for (int k = 0; k < nj; ++k) {
            std::memcpy(panel + size_t(k) * f + k, F.data() + size_t(k) * f + k, sizeof(double) * size_t(f - k));
        }
        if (nu > 0) {
            updates[t] = std::make_unique<std::vector<double>>(size_t(nu) * nu);
            double *U = updates[t]->data();
            for (int b = 0; b < nu; ++b) {
                std::memcpy(U + size_t(b) * nu + b, F.data() + size_t(nj + b) * f + nj + b, sizeof(double) * size_t(nu - b));
            }
        }
    }
    // ---- Solve schedule: fronts of equal height above the leaves are independent of one another ----
    {
        std::vector<int> height(fronts.size(), 1), where(n, -1);
        int top = 0;
        update_total = 0;
        for (int t = 0; t < int(fronts.size()); ++t) {
            auto &fr = fronts[t];
            for (const int c : fr.children) height[t] = std::max(height[t], height[c] + 1);
            top = std::max(top, height[t]);
            fr.uoff = update_total;
            update_total += fr.urows.size();
            // local index of every row this front holds, for its children's extend-add
            for (int k = 0; k < fr.nj; ++k) where[fr.j0 + k] = k;
            for (int k = 0; k < int(fr.urows.size()); ++k) where[fr.urows[k]] = fr.nj + k;
            for (const int c : fr.children) {
                auto &ch = fronts[c];
                ch.to_parent.resize(ch.urows.size());
                for (size_t i = 0; i < ch.urows.size(); ++i) ch.to_parent[i] = where[ch.urows[i]];
            }
        }
        levels.assign(size_t(top), {});
        for (int t = 0; t < int(fronts.size()); ++t) levels[size_t(height[t] - 1)].push_back(t);
    }
    return true;
}

void MultifrontalCholesky::solve(const double *b, double *x, int width) const {
    // Multifrontal solve.  Forward: every front assembles its children's update vectors into a local vector
    // (extend-add, children in stored order), eliminates its pivots and leaves its own update vector for its parent.
    // Backward: every front reads the finished unknowns at its update rows and back-substitutes its pivots.  Fronts of
    // one level (equal height above the leaves) touch disjoint data, so a level is a parallel loop; the few wide fronts
    // near the root run one at a time with the OpenMP team inside their dense kernels instead.  Every sum is formed in
    // a fixed order: the result does not depend on the team size.
    std::vector<double> y(size_t(n) * width), upd(update_total);
    for (int w = 0; w < width; ++w)
        for (int i = 0; i < n; ++i) y[size_t(w) * n + i] = b[size_t(w) * n + perm[i]];
    constexpr size_t Wide = 200000; // panel entries from which a front is worth a team of its own
    constexpr int Strip = 512;

    auto forward = [&](int t, double *yw, std::vector<double> &v, bool team) {
        const auto &fr = fronts[size_t(t)];
        const int nj = fr.nj, nu = int(fr.urows.size()), f = nj + nu;
        const double *panel = L.data() + fr.loff;
        v.assign(size_t(f), 0.0);
        for (int k = 0; k < nj; ++k) v[size_t(k)] = yw[fr.j0 + k];
        for (const int c : fr.children) {
            const auto &ch = fronts[size_t(c)];
            const double *u = upd.data() + ch.uoff;
            for (size_t i = 0; i < ch.to_parent.size(); ++i) v[size_t(ch.to_parent[i])] += u[i];
        }
        double *vp = v.data();
        for (int k = 0; k < nj; ++k) { // L11 z = v (column sweeps inside the pivot block)
            const double *col = panel + size_t(k) * f;
            const double zk = vp[k] / col[k];
            vp[k] = zk;
            if (zk == 0) continue;
            for (int i = k + 1; i < nj; ++i) vp[i] -= col[i] * zk;
        }
        const int strips = (nu + Strip - 1) / Strip; // update rows -= L21 z, row strips
#pragma omp parallel for schedule(static) if (team && strips > 1)
        for (int sidx = 0; sidx < strips; ++sidx) {
            const int i0 = nj + sidx * Strip, i1 = std::min(f, i0 + Strip);
            for (int k = 0; k < nj; ++k) {
                const double *col = panel + size_t(k) * f;
                const double zk = vp[k];
                if (zk == 0) continue;
                for (int i = i0; i < i1; ++i) vp[i] -= col[i] * zk;
            }
        }
        for (int k = 0; k < nj; ++k) yw[fr.j0 + k] = vp[k];
        std::copy(vp + nj, vp + f, upd.begin() + std::ptrdiff_t(fr.uoff));
    };
    auto backward = [&](int t, double *yw, std::vector<double> &z, bool team) {
        const auto &fr = fronts[size_t(t)];
        const int nj = fr.nj, nu = int(fr.urows.size()), f = nj + nu;
        const double *panel = L.data() + fr.loff;
        double *yj = yw + fr.j0;
        z.resize(size_t(nu));
        for (int i = 0; i < nu; ++i) z[size_t(i)] = yw[fr.urows[size_t(i)]];
        const double *zp = z.data();
#pragma omp parallel for schedule(static) if (team && nj > 64)
        for (int k = 0; k < nj; ++k) { // the update-row part of every pivot's sum: independent dot products
            const double *lower = panel + size_t(k) * f + nj;
            double s = 0;
            for (int i = 0; i < nu; ++i) s += lower[i] * zp[i];
            yj[k] -= s;
        }
        for (int k = nj - 1; k >= 0; --k) {
            const double *col = panel + size_t(k) * f;
            double s = yj[k];
            for (int i = k + 1; i < nj; ++i) s -= col[i] * yj[i];
            yj[k] = s / col[k];
        }
    };
    auto sweep = [&](const std::vector<int> &level, double *yw, auto &&kernel) {
        size_t widest = 0;
        for (const int t : level) widest = std::max(widest, (fronts[size_t(t)].urows.size() + size_t(fronts[size_t(t)].nj)) * size_t(fronts[size_t(t)].nj));
        if (level.size() < 4 || widest >= 16 * Wide) { // near the root: one front at a time, the team inside it
            std::vector<double> scratch;
            for (const int t : level) kernel(t, yw, scratch, (fronts[size_t(t)].urows.size() + size_t(fronts[size_t(t)].nj)) * size_t(fronts[size_t(t)].nj) >= Wide);
            return;
        }
#pragma omp parallel
        {
            std::vector<double> scratch;
#pragma omp for schedule(dynamic, 1)
            for (int idx = 0; idx < int(level.size()); ++idx) kernel(level[size_t(idx)], yw, scratch, false);
        }
    };
    for (int w = 0; w < width; ++w) {
        double *yw = y.data() + size_t(w) * n;
        for (size_t h = 0; h < levels.size(); ++h) sweep(levels[h], yw, forward);
        for (size_t h = levels.size(); h-- > 0;) sweep(levels[h], yw, backward);
    }
    for (int w = 0; w < width; ++w)
        for (int i = 0; i < n; ++i) x[size_t(w) * n + perm[i]] = y[size_t(w) * n + i];
}
} // namespace oracle
