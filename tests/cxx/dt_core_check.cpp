// CPU check of csrc/dt_core.h -- the leaf / merge code k_delaunay runs, compiled for the host over plain memory and
// driven exactly as the kernel drives it (alternating-cut order, bottom-up by depth, record ranges known in advance).
// The triangle lists must equal csrc/delaunay.cpp's (order included), which the oracle tests pin against the real
// Triangle.  Both record storages (MeshG: 32-bit rows, MeshL: 24-byte records) and both forms of the seam step
// (kShort on / off) are run; with kShort every shortcut is compared with the fresh read it replaces.
//   usage: dt_core_check [rounds] [seed]        exit 0 = identical everywhere
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

static long g_short_taken = 0, g_short_bad = 0;
#define DT_CHECK_SHORTCUT(cond) do { g_short_taken++; if (!(cond)) g_short_bad++; } while (0)
#include "../../stereo-vision_amd/csrc/dt_core.h"
#include "../../stereo-vision_amd/csrc/svh_internal.h"

using namespace svh;
using namespace svh::dt;

struct Pt { int x, y, id; };

// alternating-cut order of distinct points (triangle.cpp:5582-5604, 6198-6206): lower half by (x, y) at the root,
// then the axes alternate; subsets of <= 3 are ordered by (x, y)
static void kd(Pt* a, int n, int axis) {
    auto by_x = [](const Pt& p, const Pt& q) { return p.x < q.x || (p.x == q.x && p.y < q.y); };
    auto by_y = [](const Pt& p, const Pt& q) { return p.y < q.y || (p.y == q.y && p.x < q.x); };
    if (n <= 3) { std::sort(a, a + n, by_x); return; }
    if (axis == 0) std::sort(a, a + n, by_x); else std::sort(a, a + n, by_y);
    const int h = n >> 1;
    kd(a, h, 1 - axis);
    kd(a + h, n - h, 1 - axis);
}

template <bool kShort, class M, class IdsOf>
static std::vector<int32_t> run_mesh(const M& mesh, const std::vector<Pt>& P, IdsOf ids_of) {
    const int m = (int)P.size();
    std::vector<int> order(m), oxy(m);
    for (int i = 0; i < m; i++) { order[i] = P[i].id; oxy[i] = P[i].x | P[i].y << 16; }
    int depth = 0;
    for (;; depth++) {
        bool split = false;
        for (int i = 0; i < m && !split; i++) { int s, n; dt_segment(m, depth, i, &s, &n); split = n > 3; }
        if (!split) break;
    }
    std::vector<unsigned> FL(2 * (size_t)m), FR(2 * (size_t)m);
    mesh.make_rec(0);
    for (int d = depth; d >= 0; d--) {
        unsigned* fl = FL.data() + (size_t)(d & 1) * m;
        unsigned* fr = FR.data() + (size_t)(d & 1) * m;
        const unsigned* cfl = FL.data() + (size_t)((d + 1) & 1) * m;
        const unsigned* cfr = FR.data() + (size_t)((d + 1) & 1) * m;
        for (unsigned j = 0; j < (1u << d); j++) dt_node<kShort>(mesh, m, d, j, order.data(), oxy.data(), cfl, cfr, fl, fr);
    }
    std::vector<int32_t> out;
    for (int t = 1; t < 2 * m - 1; t++) {
        int v[3];
        ids_of(t, v);
        if (v[0] >= 0 && v[1] >= 0 && v[2] >= 0) { out.push_back(v[1]); out.push_back(v[2]); out.push_back(v[0]); }
    }
    return out;
}

static std::vector<Pt> make_points(std::mt19937& rng, int kind, int n) {
    std::vector<Pt> p;
    auto U = [&](int hi) { return (int)(rng() % (unsigned)hi); };
    for (int i = 0; i < n; i++) {
        int x, y;
        switch (kind) {
            case 0: x = 5 * U(300); y = 5 * U(75); break;                    // support lattice: co-circular quads everywhere
            case 1: x = U(1500); y = U(375); break;                          // integer pixels
            case 2: x = 5 * U(40); y = 5 * U(3); break;                      // three lattice rows: long collinear runs
            case 3: x = 5 * U(200); y = 100; break;                          // ONE row: every subset is collinear
            case 4: x = 300; y = U(370); break;                              // one column
            case 5: x = U(12); y = U(12); break;                             // tiny dense grid
            case 6: x = 5 * U(300); y = (U(4) == 0) ? 5 * U(75) : 5 * (U(3) + 10); break;   // a dense band + sparse rest
            default: x = U(16000); y = U(16000); break;                      // the whole coordinate range
        }
        p.push_back({x, y, 0});
    }
    // distinct points only (the kernel drops coincident ones before the build)
    std::sort(p.begin(), p.end(), [](const Pt& a, const Pt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    p.erase(std::unique(p.begin(), p.end(), [](const Pt& a, const Pt& b) { return a.x == b.x && a.y == b.y; }), p.end());
    std::shuffle(p.begin(), p.end(), rng);
    for (size_t i = 0; i < p.size(); i++) p[i].id = (int)i;
    return p;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 20260929u);
    long sets = 0, tris = 0, bad = 0;
    for (int r = 0; r < rounds; r++) {
        const int kind = r % 8;
        const int nmax = kind == 5 ? 140 : (kind == 3 || kind == 4 ? 200 : (r % 3 == 0 ? 40 : 3000));
        std::vector<Pt> P = make_points(rng, kind, 2 + (int)(rng() % (unsigned)nmax));
        const int m = (int)P.size();
        if (m < 2) continue;
        // expected: csrc/delaunay.cpp on the same points (input order = ids)
        std::vector<float> pts(2 * (size_t)m);
        for (const Pt& p : P) { pts[2 * p.id] = (float)p.x; pts[2 * p.id + 1] = (float)p.y; }
        std::vector<int32_t> want(3 * (2 * (size_t)m + 16));
        const int32_t nt = delaunay(pts.data(), m, want.data(), 2 * m + 16, 0);
        want.resize(3 * (size_t)std::max(nt, 0));
        std::vector<Pt> Q = P;
        kd(Q.data(), m, 0);
        const size_t nrec = 2 * (size_t)m + 2;
        for (int form = 0; form < 4; form++) {
            std::vector<int32_t> got;
            if (form < 2) {
                std::vector<int> ids(4 * nrec, 0x5a5a5a5a), xys(4 * nrec, 0x5a5a5a5a);
                std::vector<unsigned> nbr(4 * nrec, 0x5a5a5a5au);
                MeshG g{ids.data(), xys.data(), nbr.data()};
                auto ids_of = [&](int t, int v[3]) { v[0] = ids[4 * t]; v[1] = ids[4 * t + 1]; v[2] = ids[4 * t + 2]; };
                got = form == 0 ? run_mesh<false>(g, Q, ids_of) : run_mesh<true>(g, Q, ids_of);
            } else {
                if (m > 8000) continue;
                std::vector<uint64_t> mem(3 * nrec, 0x5a5a5a5a5a5a5a5aull);
                MeshL l{reinterpret_cast<unsigned char*>(mem.data())};
                auto ids_of = [&](int t, int v[3]) {
                    const Rec rc = l.load((unsigned)t * 4u);
                    v[0] = rc.id0; v[1] = rc.id1; v[2] = rc.id2;
                };
                got = form == 2 ? run_mesh<false>(l, Q, ids_of) : run_mesh<true>(l, Q, ids_of);
            }
            if (got != want) {
                bad++;
                if (bad <= 10)
                    fprintf(stderr, "MISMATCH round %d kind %d m %d form %d: %zu vs %zu triangles\n", r, kind, m, form,
                            got.size() / 3, want.size() / 3);
            }
        }
        sets++;
        tris += nt;
    }
    printf("dt_core_check: %ld point sets x 4 forms (two storages, shortcut on / off), %ld triangles, mismatches %ld, shortcuts taken %ld, shortcuts wrong %ld\n",
           sets, tris, bad, g_short_taken, g_short_bad);
    return bad || g_short_bad ? 1 : 0;
}
