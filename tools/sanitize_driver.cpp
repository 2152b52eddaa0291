// Sanitizer driver for the threaded host code of libsvhip: the parallel
// divide-and-conquer of delaunay.cpp with its parked helper threads (run_pair / HelperPool spin
// on atomics), and the lattice filters of elas_host.cpp, both hammered from several caller threads
// at once the way the batch workers and the single-call path use them.
//   make -C stereo-vision_amd sanitize     (builds this with -fsanitize=thread and with
//                                           -fsanitize=address,undefined and runs both)
// Exits non-zero when a parallel triangulation differs from the sequential one.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <random>
#include <thread>
#include <vector>

#include "../stereo-vision_amd/csrc/svh_internal.h"

using namespace svh;

static std::vector<float> make_points(std::mt19937& rng, int kind, int n) {
    std::vector<float> p;
    auto U = [&](int hi) { return (int)(rng() % (unsigned)hi); };
    for (int i = 0; i < n; i++) {
        float x, y;
        switch (kind) {
            case 0: x = 5.f * U(240); y = 5.f * U(70); break;                 // lattice: co-circular quads
            case 1: x = (float)U(1200); y = (float)U(370); break;             // integer pixels
            case 2: x = U(400) * 0.5f; y = U(200) * 0.25f; break;             // dyadic fractions (128-bit path)
            case 3: x = 5.f * U(40); y = 5.f * U(3); break;                   // many duplicates, collinear runs
            default: x = (float)U(30000) - 15000.f; y = (float)U(30000) - 15000.f; break;   // large, negative
        }
        p.push_back(x);
        p.push_back(y);
    }
    return p;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    std::atomic<int> bad{0};
    std::atomic<long> tris{0}, sups{0};
    auto worker = [&](int id) {
        std::mt19937 rng(1234 + id);
        for (int r = 0; r < rounds; r++) {
            // ---- triangulations: sequential vs parallel depths, same output required
            const int kind = (r + id) % 5, n = 3 + (int)(rng() % (kind == 3 ? 300 : 2500));
            std::vector<float> pts = make_points(rng, kind, n);
            std::vector<int32_t> a(3 * (2 * n + 16)), b(a.size());
            const int32_t na = delaunay(pts.data(), n, a.data(), 2 * n + 16, 0);
            for (int depth = 1; depth <= 3; depth++) {
                // (odd depths with the Matcher's hint: the mirrored quicksort without the radix sort ahead of it)
                const int32_t nb = delaunay(pts.data(), n, b.data(), 2 * n + 16, depth, (depth & 1) != 0);
                if (na != nb || (na > 0 && memcmp(a.data(), b.data(), sizeof(int32_t) * 3 * na) != 0)) {
                    fprintf(stderr, "thread %d round %d kind %d n %d depth %d: parallel result differs\n", id, r, kind,
                            n, depth);
                    bad++;
                }
            }
            if (na > 0) tris += na;
            // ---- the helper calls of the Matcher's vote: a warm window, then a four-way split with private outputs
            if (r % 3 == 0) helpers_warm(3, 200);
            long part[4] = {0, 0, 0, 0};
            run_many(4, [&](int i) {
                for (int32_t t = na * i / 4; t < na * (i + 1) / 4; t++) part[i] += a[3 * t] + a[3 * t + 1] + a[3 * t + 2];
            });
            long whole = 0;
            for (int32_t t = 0; t < 3 * std::max(na, 0); t++) whole += a[t];
            if (part[0] + part[1] + part[2] + part[3] != whole) {
                fprintf(stderr, "thread %d round %d: run_many lost work\n", id, r);
                bad++;
            }
            if (r % 3 == 0) helpers_warm(0, 0);
            // ---- both triangulations of a support list on two threads (single-call path)
            HostPrior hp;
            for (int i = 0; i < 400; i++) {
                const int u = 5 * (1 + (int)(rng() % 240)), v = 5 * (1 + (int)(rng() % 70)), d = (int)(rng() % 60);
                if (u - d < 0) continue;
                hp.support.push_back(u);
                hp.support.push_back(v);
                hp.support.push_back(d);
            }
            if (!triangulate_support(hp, /*parallel=*/true)) bad++;
            // ---- lattice filters on a random candidate lattice (SSE2 form and scalar form)
            svh_elas_params p;
            svh_elas_params_default(&p, r & 1);
            if (r % 7 == 3) p.incon_window_size = 9;   // scalar form
            const Dims d = make_dims(p, 1242, 375);
            std::vector<int16_t> dc((size_t)d.Wc * d.Hc);
            for (auto& x : dc) x = (rng() % 3) ? (int16_t)-1 : (int16_t)(20 + rng() % 6 + (rng() % 11 == 0 ? 40 : 0));
            for (int i = 0; i < d.Wc; i++) dc[i] = 0;
            for (int j = 0; j < d.Hc; j++) dc[(size_t)j * d.Wc] = 0;
            std::vector<int32_t> s1, s2;
            std::vector<int16_t> c2 = dc;
            support_from_candidates(p, d, dc.data(), s1, /*write_back=*/false);
            support_from_candidates(p, d, c2.data(), s2, /*write_back=*/true);
            if (s1 != s2) bad++;
            sups += (long)s1.size() / 3;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < 6; t++) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
    printf("sanitize_driver: 6 threads x %d rounds, %ld triangles, %ld support points, mismatches %d\n", rounds,
           tris.load(), sups.load(), bad.load());
    return bad.load() ? 1 : 0;
}

// the one symbol elas_host.cpp / the driver need from the engine
extern "C" void svh_elas_params_default(svh_elas_params* p, int32_t setting) {
    const bool rob = setting == SVH_ELAS_ROBOTICS;
    p->disp_min = 0; p->disp_max = 255; p->support_threshold = rob ? 0.85f : 0.95f; p->support_texture = 10;
    p->candidate_stepsize = 5; p->incon_window_size = 5; p->incon_threshold = 5; p->incon_min_support = 5;
    p->add_corners = rob ? 0 : 1; p->grid_size = 20; p->beta = 0.02f; p->gamma = rob ? 3.f : 5.f; p->sigma = 1.f;
    p->sradius = rob ? 2.f : 3.f; p->match_texture = rob ? 1 : 0; p->lr_threshold = 2; p->speckle_sim_threshold = 1.f;
    p->speckle_size = 200; p->ipol_gap_width = rob ? 3 : 5000; p->filter_median = rob ? 0 : 1;
    p->filter_adaptive_mean = rob ? 1 : 0; p->postprocess_only_left = rob ? 1 : 0; p->subsampling = 0;
}
