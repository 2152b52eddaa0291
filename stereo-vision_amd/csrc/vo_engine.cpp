// VisualOdometryStereo behind the C-ABI (include/svh.h, svh_vo_*)
//   libviso2/src/viso_stereo.cpp:26-68 (constructor, process)
//   libviso2/src/viso.cpp:28-64 (constructor, updateMotion), :68-96, :130-153
// The Matcher calls go through the svh_matcher_* entries of this library; the motion
// estimate runs in two kernels (vo_kernels.hip).  There is no CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/svh.h"
#include "batch_rec.h"
#include "vo_internal.h"

using namespace svh;

namespace svh {
int fail(int code, const std::string& msg);   // elas_engine.cpp: records svh_last_error()
bool fi_armed();                                    // elas_engine.cpp: fault injection (svh_internal.h)
bool fi_hit(const char* expr_text);
void report_hip_failure(const char* entry);
}

static int vo_hip_failed(const char* expr, bool injected, hipError_t e) {
    const int rc = svh::fail(SVH_ERR_HIP, std::string(expr) + ": " + (injected ? "injected failure (SVH_TEST_FAIL_AT)" : hipGetErrorString(e)));
    svh::report_hip_failure("VisualOdometry");
    return rc;
}
#define VO_TRY(expr)                                                                             \
    do {                                                                                         \
        const bool inj_ = svh::fi_armed() && svh::fi_hit(#expr);   /* svh_internal.h: fault injection */ \
        hipError_t e_ = inj_ ? hipErrorUnknown : (expr);                                         \
        if (e_ != hipSuccess) return vo_hip_failed(#expr, inj_, e_);                             \
    } while (0)

struct svh_vo {
    svh_vo_params p;
    svh_matcher* matcher = nullptr;
    double Tr[16];
    bool Tr_valid = false;
    RandStream rng;   // libc rand() unless svh_vo_set_private_rand switched the object to its own stream
    std::vector<int32_t> inliers;
    std::vector<svh_p_match> matched;
    int device = 0;
    hipStream_t stream = nullptr;
    // pinned host staging: [matches | samples] in, VoResult + inlier list out
    uint8_t* h_in = nullptr;
    size_t h_in_cap = 0;
    VoResult* h_out = nullptr;
    int32_t* h_inl = nullptr;
    int32_t inl_cap = 0;
    // device scratch
    uint8_t* d_in = nullptr;
    size_t d_in_cap = 0;
    double* d_hyp_tr = nullptr;
    int32_t* d_hyp_count = nullptr;
    int32_t hyp_cap = 0;
    uint8_t* d_flags = nullptr;
    size_t flags_cap = 0;
    double* d_J = nullptr;
    double* d_res = nullptr;
    int32_t j_cap = 0;
};

namespace {

template <typename T>
hipError_t grow(T** p, size_t n) {
    (void)hipFree(*p);
    *p = nullptr;
    return hipMalloc((void**)p, n * sizeof(T));
}

int ensure(svh_vo* v, int32_t N, int32_t iters) {
    if (!v->stream) {
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0)
            return svh::fail(SVH_ERR_NO_DEVICE, "no HIP device visible: libsvhip has no CPU fallback");
        VO_TRY(hipSetDevice(v->device));
        VO_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    }
    VO_TRY(hipSetDevice(v->device));
    if (!v->h_out) VO_TRY(hipHostMalloc((void**)&v->h_out, sizeof(VoResult)));   // (its own check: a failure here must not be hidden behind the stream's)
    const size_t in_bytes = (((size_t)N * sizeof(svh_p_match) + 15) & ~(size_t)15) +
                            (((size_t)iters * 3 * sizeof(int32_t) + 15) & ~(size_t)15);
    // (every capacity goes to 0 BEFORE its buffer is freed and back up only when the new one exists: a failed
    // allocation must not leave a freed pointer behind a capacity that says "fits" to a later, smaller request)
    if (in_bytes > v->h_in_cap) {
        v->h_in_cap = 0;
        (void)hipHostFree(v->h_in);
        v->h_in = nullptr;
        VO_TRY(hipHostMalloc((void**)&v->h_in, in_bytes));
        v->h_in_cap = in_bytes;
    }
    if (in_bytes > v->d_in_cap) {
        v->d_in_cap = 0;
        VO_TRY(grow(&v->d_in, in_bytes));
        v->d_in_cap = in_bytes;
    }
    if (N > v->inl_cap) {
        v->inl_cap = 0;
        (void)hipHostFree(v->h_inl);
        v->h_inl = nullptr;
        VO_TRY(hipHostMalloc((void**)&v->h_inl, (size_t)N * sizeof(int32_t)));
        v->inl_cap = N;
    }
    if (iters > v->hyp_cap) {
        v->hyp_cap = 0;
        VO_TRY(grow(&v->d_hyp_tr, (size_t)6 * iters));
        VO_TRY(grow(&v->d_hyp_count, (size_t)iters));
        v->hyp_cap = iters;
    }
    if ((size_t)iters * N > v->flags_cap) {
        v->flags_cap = 0;
        VO_TRY(grow(&v->d_flags, (size_t)iters * N));
        v->flags_cap = (size_t)iters * N;
    }
    if (N > v->j_cap) {
        v->j_cap = 0;
        VO_TRY(grow(&v->d_J, (size_t)24 * N));
        VO_TRY(grow(&v->d_res, (size_t)4 * N));
        v->j_cap = N;
    }
    return SVH_OK;
}

// Matrix VisualOdometry::transformationVectorToMatrix   viso.cpp:68-96
void vector_to_matrix(const double* tr, double* T) {
    const double sx = sin(tr[0]), cx = cos(tr[0]), sy = sin(tr[1]), cy = cos(tr[1]);
    const double sz = sin(tr[2]), cz = cos(tr[2]);
    T[0] = +cy * cz;                T[1] = -cy * sz;                T[2] = +sy;       T[3] = tr[3];
    T[4] = +sx * sy * cz + cx * sz; T[5] = -sx * sy * sz + cx * cz; T[6] = -sx * cy;  T[7] = tr[4];
    T[8] = -cx * sy * cz + sx * sz; T[9] = +cx * sy * sz + sx * cz; T[10] = +cx * cy; T[11] = tr[5];
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// estimateMotion in three steps (a batched call interleaves them over its objects):
//   estimate_prepare: buffers, matches and the RANSAC samples into pinned memory (consumes libc rand())
//   estimate_enqueue: upload + the two kernels;   estimate_collect: result after the stream wait
// estimate_prepare returns 1 when there is device work, 0 for the reference's early return (N < 6), <0 on error
int estimate_prepare(svh_vo* v, const svh_p_match* pm, int32_t N) {
    const svh_vo_params& P = v->p;
    if (N < 6) return 0;   // viso_stereo.cpp:91-94: returns before _inliers is cleared
    const int32_t iters = P.ransac_iters > 0 ? P.ransac_iters : 0;
    int rc = ensure(v, N, iters);
    if (rc) return rc;
    v->inliers.clear();
    // getRandomSample(N, 3) for every iteration, viso.cpp:130-153: three libc rand() draws
    // without replacement (the draw indexes the list of the not yet chosen indices)
    const size_t m_bytes = ((size_t)N * sizeof(svh_p_match) + 15) & ~(size_t)15;
    memcpy(v->h_in, pm, (size_t)N * sizeof(svh_p_match));
    int32_t* samples = reinterpret_cast<int32_t*>(v->h_in + m_bytes);
    for (int32_t k = 0; k < iters; k++) {
        int32_t chosen[3];
        for (int q = 0; q < 3; q++) {
            int32_t j = v->rng.next() % (N - q);
            // j-th smallest index not chosen yet
            int32_t idx = j;
            bool moved = true;
            while (moved) {
                moved = false;
                int32_t below = 0;
                for (int r = 0; r < q; r++) below += chosen[r] <= idx;
                if (idx != j + below) {
                    idx = j + below;
                    moved = true;
                }
            }
            chosen[q] = idx;
            samples[3 * k + q] = idx;
        }
    }
    return 1;
}

void estimate_enqueue(svh_vo* v, int32_t N) {
    const svh_vo_params& P = v->p;
    const int32_t iters = P.ransac_iters > 0 ? P.ransac_iters : 0;
    const size_t m_bytes = ((size_t)N * sizeof(svh_p_match) + 15) & ~(size_t)15;
    const size_t s_bytes = ((size_t)iters * 3 * sizeof(int32_t) + 15) & ~(size_t)15;
    hipStream_t s = v->stream;
    vlaunch_upload(s, v->h_in, v->d_in, m_bytes + s_bytes);
    VoCalib c;
    c.f = P.f; c.cu = P.cu; c.cv = P.cv; c.base = P.base;
    c.inlier_threshold = P.inlier_threshold;
    c.reweighting = P.reweighting;
    vlaunch_estimate(s, reinterpret_cast<const svh_p_match*>(v->d_in), N,
                     reinterpret_cast<const int32_t*>(v->d_in + m_bytes), iters, c, v->d_hyp_tr,
                     v->d_hyp_count, v->d_flags, v->d_J, v->d_res, v->h_out, v->h_inl);
}

int estimate_collect(svh_vo* v, double* tr6) {
    const VoResult& r = *v->h_out;
    v->inliers.assign(v->h_inl, v->h_inl + r.n_inliers);
    if (!r.success) return 0;
    for (int i = 0; i < 6; i++) tr6[i] = r.tr[i];
    return 1;
}

// estimateMotion: returns 1 + tr, 0 for the reference's empty vector, <0 on error
int estimate(svh_vo* v, const svh_p_match* pm, int32_t N, double* tr6) {
    const int rc = estimate_prepare(v, pm, N);
    if (rc <= 0) return rc;
    estimate_enqueue(v, N);
    VO_TRY((hipError_t)wait_stream(v->stream));
    VO_TRY(hipGetLastError());
    return estimate_collect(v, tr6);
}

// bool VisualOdometry::updateMotion   viso.cpp:47-64
int update_motion(svh_vo* v) {
    double tr[6];
    const int ok = estimate(v, v->matched.data(), (int32_t)v->matched.size(), tr);
    if (ok <= 0) return ok;
    vector_to_matrix(tr, v->Tr);
    v->Tr_valid = true;
    return 1;
}

void fetch_matches(svh_vo* v) {
    const int32_t n = svh_matcher_get_matches(v->matcher, nullptr, 0);
    v->matched.resize(n);
    if (n) svh_matcher_get_matches(v->matcher, v->matched.data(), n);
}

}  // namespace

extern "C" {

void svh_vo_params_default(svh_vo_params* p) {
    if (!p) return;
    svh_matcher_params_default(&p->match);
    p->bucket_max_features = 2;      // viso.h:50-55
    p->bucket_width = 50;
    p->bucket_height = 50;
    p->f = 1; p->cu = 0; p->cv = 0;  // viso.h:38-43
    p->base = 1.0;                   // viso_stereo.h:38-43
    p->ransac_iters = 200;
    p->inlier_threshold = 2.0;
    p->reweighting = 1;
}

svh_vo* svh_vo_create(const svh_vo_params* p) {
    svh::ensure_init();
    if (!p) return nullptr;
    svh_vo* v = new svh_vo();
    v->p = *p;
    v->matcher = svh_matcher_create(&p->match);
    if (!v->matcher) {
        delete v;
        return nullptr;
    }
    svh_matcher_set_intrinsics(v->matcher, p->f, p->cu, p->cv, p->base);   // viso_stereo.cpp:30
    for (int i = 0; i < 16; i++) v->Tr[i] = (i % 5 == 0) ? 1.0 : 0.0;       // Matrix::eye(4)
    // The HIP runtime draws from libc rand() while it initialises (first allocation, code
    // object load).  Do all of that BEFORE the reference's srand(0), so that afterwards the
    // rand() stream is consumed by bucketFeatures / getRandomSample only, as in the reference.
    int nd = 0;
    if (hipGetDeviceCount(&nd) == hipSuccess && nd > 0) {
        (void)hipGetDevice(&v->device);
        (void)ensure(v, 64, 8);
        svh_p_match dummy[8];
        memset(dummy, 0, sizeof(dummy));
        double tr[6];
        (void)estimate(v, dummy, 8, tr);           // launches (and so loads) the kernels
        v->inliers.clear();
        const int32_t wd[3] = {64, 32, 64};
        std::vector<uint8_t> blank((size_t)64 * 32, 0);
        svh_matcher* tmp = svh_matcher_create(&p->match);     // a scratch Matcher: streams, kernels
        (void)svh_matcher_push_back(tmp, blank.data(), blank.data(), wd, 0);
        (void)svh_matcher_push_back(tmp, blank.data(), blank.data(), wd, 0);
        (void)svh_matcher_match_features(tmp, 2, nullptr);
        svh_matcher_destroy(tmp);
    }
    srand(0);                                                               // viso.cpp:36
    return v;
}

void svh_vo_destroy(svh_vo* v) {
    if (!v) return;
    if (v->stream) {
        (void)hipSetDevice(v->device);
        (void)hipStreamSynchronize(v->stream);
        (void)hipFree(v->d_in); (void)hipFree(v->d_hyp_tr); (void)hipFree(v->d_hyp_count);
        (void)hipFree(v->d_flags); (void)hipFree(v->d_J); (void)hipFree(v->d_res);
        (void)hipHostFree(v->h_in); (void)hipHostFree(v->h_out); (void)hipHostFree(v->h_inl);
        (void)hipStreamDestroy(v->stream);
    }
    svh_matcher_destroy(v->matcher);
    delete v;
}

// VisualOdometryStereo::process after its pushBack   viso_stereo.cpp:47-68
static int32_t process_after_push(svh_vo* v) {
    const svh_vo_params& P = v->p;
    int32_t rc;
    if (!v->Tr_valid) {   // bootstrap (viso_stereo.cpp:47-53)
        rc = svh_matcher_match_features(v->matcher, 2, nullptr);
        if (rc < 0) return rc;
        bucket_features(v->matcher, P.bucket_max_features, (float)P.bucket_width, (float)P.bucket_height, v->rng);
        fetch_matches(v);
        rc = update_motion(v);
        if (rc < 0) return rc;
    }
    rc = svh_matcher_match_features(v->matcher, 2, v->Tr_valid ? v->Tr : nullptr);
    if (rc < 0) return rc;
    bucket_features(v->matcher, P.bucket_max_features, (float)P.bucket_width, (float)P.bucket_height, v->rng);
    fetch_matches(v);
    return update_motion(v);
}

int32_t svh_vo_process(svh_vo* v, const uint8_t* I1, const uint8_t* I2, const int32_t* dims, int32_t replace) {
    svh::ActiveCaller active_;
    if (!v || !dims) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    const int32_t rc = svh_matcher_push_back(v->matcher, I1, I2, dims, replace);
    if (rc < 0 && rc != SVH_ERR_BAD_DIMS) return rc;   // (bad dimensions: message printed, frame ignored -- viso_stereo.cpp:41-68 goes on; a missing or pending prefetched frame IS an error)   // bad dims: message printed, carry on like the reference
    return process_after_push(v);
}

// One frame of K sequences: K VisualOdometryStereo objects in lockstep.  The Matcher steps go through the batched
// Matcher entries, the K motion estimates are two launches.  libc rand() is consumed in the order of K
// svh_vo_process calls (object by object: bucketing, then the RANSAC samples), so with the same srand the results
// are bit-identical to that loop.  ok[i] (optional) = what svh_vo_process would have returned for object i
// (1 motion updated, 0 estimate failed); the call returns <0 on the first error, else the number of successes.
// Objects that are still bootstrapping (no valid motion yet), or that differ in parameters, make the call run
// them one after the other.
static int32_t process_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* I1, const uint8_t* const* I2,
                             const int32_t* dims, int32_t replace, int32_t* ok, const uint8_t* const* N1,
                             const uint8_t* const* N2) {
    svh::ActiveCaller active_;
    // I1 == I2 == NULL: the objects take the frame handed over by svh_vo_prefetch_batch
    if (!vs || K < 0 || (!I1 != !I2) || !dims) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    bool lockstep = K > 1;
    for (int i = 0; i < K; i++) {
        if (!vs[i]) return svh::fail(SVH_ERR_BAD_ARG, "null object in the batch");
        for (int j = 0; j < i; j++)
            if (vs[j] == vs[i]) return svh::fail(SVH_ERR_BAD_ARG, "the same object twice in one batch");
        lockstep = lockstep && vs[i]->Tr_valid && memcmp(&vs[i]->p, &vs[0]->p, sizeof(vs[0]->p)) == 0 &&
                   vs[i]->device == vs[0]->device;
    }
    int32_t good = 0;
    std::vector<svh_matcher*> ms(K);
    for (int i = 0; i < K; i++) ms[i] = vs[i]->matcher;
    if (!lockstep) {
        // one by one, in the order of K svh_vo_process calls (pushBack draws no random numbers, so taking all
        // K frames first -- which the hand-over of the next frame needs -- keeps the draw order)
        for (int i = 0; i < K; i++) {
            const int32_t rc = svh_matcher_push_back(ms[i], I1 ? I1[i] : nullptr, I2 ? I2[i] : nullptr, dims, replace);
            if (rc < 0 && rc != SVH_ERR_BAD_DIMS) return rc;   // (bad dimensions: message printed, frame ignored -- viso_stereo.cpp:41-68 goes on; a missing or pending prefetched frame IS an error)
        }
        if (N1) {
            const int32_t rc = svh_matcher_prefetch_batch(ms.data(), K, N1, N2, dims);
            if (rc < 0) return rc;
        }
        for (int i = 0; i < K; i++) {
            const int32_t rc = process_after_push(vs[i]);
            if (rc < 0) return rc;
            if (ok) ok[i] = rc;
            good += rc > 0;
        }
        return good;
    }
    static const bool timing = svh::env("SVH_MATCHER_TIMING") != nullptr;
    struct Acc {
        double t[5] = {0, 0, 0, 0, 0};
        int64_t calls = 0;
        ~Acc() {
            if (calls)
                fprintf(stderr, "[svh lockstep timing] vo: pushBack %.3f, matchFeatures %.3f, bucketing+samples %.3f, "
                                "estimate %.3f ms per call\n", t[0] / calls, t[1] / calls, t[2] / calls, t[3] / calls);
        }
    };
    static Acc acc;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt[5] = {0, 0, 0, 0, 0};
    if (timing) tt[0] = now();
    const svh_vo_params& P = vs[0]->p;
    std::vector<const double*> trs(K);
    for (int i = 0; i < K; i++) trs[i] = vs[i]->Tr;
    int32_t rc = svh_matcher_push_back_batch(ms.data(), K, I1, I2, dims, replace);
    if (rc < 0 && rc != SVH_ERR_BAD_DIMS) return rc;   // (bad dimensions: message printed, frame ignored -- viso_stereo.cpp:41-68 goes on; a missing or pending prefetched frame IS an error)
    if (N1) {   // the next frame goes out now: its packing, upload and features overlap everything below
        rc = svh_matcher_prefetch_batch(ms.data(), K, N1, N2, dims);
        if (rc < 0) return rc;
    }
    if (timing) tt[1] = now();
    rc = svh_matcher_match_features_batch(ms.data(), K, 2, trs.data());
    if (rc < 0) return rc;
    if (timing) tt[2] = now();
    std::vector<int> state(K, 0);
    auto select = [&](int i) {
        svh_vo* v = vs[i];
        bucket_features(v->matcher, P.bucket_max_features, (float)P.bucket_width, (float)P.bucket_height, v->rng);
        fetch_matches(v);
        state[i] = estimate_prepare(v, v->matched.data(), (int32_t)v->matched.size());
    };
    bool all_private = true;
    for (int i = 0; i < K; i++) all_private = all_private && vs[i]->rng.is_private;
    if (all_private) {
        // every object draws from its own stream: no order to keep between them
        batch_parallel_for(K, [&](int i) {
            (void)hipSetDevice(vs[i]->device);
            select(i);
        });
    } else {
        for (int i = 0; i < K; i++) select(i);   // the process-wide rand(): in the order of K svh_vo_process calls
    }
    for (int i = 0; i < K; i++)
        if (state[i] < 0) return state[i];
    if (timing) tt[3] = now();
    BatchRec& rec = batch_recorder(vs[0]->device);
    rec.reset();
    t_rec = &rec;
    for (int i = 0; i < K; i++)
        if (state[i] > 0) {
            rec.begin_object();
            estimate_enqueue(vs[i], (int32_t)vs[i]->matched.size());
        }
    t_rec = nullptr;
    if (rec.broken) {
        rec.reset();
        for (int i = 0; i < K; i++)
            if (state[i] > 0) {
                estimate_enqueue(vs[i], (int32_t)vs[i]->matched.size());
                VO_TRY((hipError_t)wait_stream(vs[i]->stream));
            }
    } else {
        hipStream_t s = vs[0]->stream;
        VO_TRY(rec.flush(s));
        VO_TRY((hipError_t)wait_stream(s));
        rec.synced();
    }
    VO_TRY(hipGetLastError());
    for (int i = 0; i < K; i++) {
        int32_t r = 0;
        if (state[i] > 0) {
            double tr[6];
            r = estimate_collect(vs[i], tr);
            if (r > 0) {
                vector_to_matrix(tr, vs[i]->Tr);
                vs[i]->Tr_valid = true;
            }
        }
        if (ok) ok[i] = r;
        good += r > 0;
    }
    if (timing) {
        tt[4] = now();
        for (int i = 0; i < 4; i++) acc.t[i] += tt[i + 1] - tt[i];
        acc.calls++;
    }
    return good;
}

int32_t svh_vo_process_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* I1, const uint8_t* const* I2,
                             const int32_t* dims, int32_t replace, int32_t* ok) {
    return process_batch(vs, K, I1, I2, dims, replace, ok, nullptr, nullptr);
}

// The pipelined form: processes the frame handed over before (svh_vo_prefetch_batch, or the `next` images of the
// previous call) and hands over the next one (next_I1 / next_I2, or NULL at the end of the sequence) right after
// the ring buffers rotated, so that its packing, upload and feature extraction overlap this frame's matching and
// motion estimate.  Results are those of svh_vo_process_batch with the images passed directly.
int32_t svh_vo_process_next_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* next_I1,
                                  const uint8_t* const* next_I2, const int32_t* dims, int32_t replace, int32_t* ok) {
    if ((!next_I1) != (!next_I2)) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    return process_batch(vs, K, nullptr, nullptr, dims, replace, ok, next_I1, next_I2);
}

// The NEXT frame of the K objects handed over early (svh_matcher_prefetch_batch).  Per iteration: prefetch frame
// t+1, then svh_vo_process_batch without images for frame t (prefetched in the iteration before): the packing,
// upload and feature extraction of t+1 overlap the matching and the motion estimate of t.
int32_t svh_vo_prefetch_batch(svh_vo* const* vs, int32_t K, const uint8_t* const* I1, const uint8_t* const* I2,
                              const int32_t* dims) {
    if (!vs || K <= 0 || !I1 || !I2 || !dims) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    std::vector<svh_matcher*> ms(K);
    for (int i = 0; i < K; i++) {
        if (!vs[i]) return svh::fail(SVH_ERR_BAD_ARG, "null object in the batch");
        ms[i] = vs[i]->matcher;
    }
    return svh_matcher_prefetch_batch(ms.data(), K, I1, I2, dims);
}

int32_t svh_vo_process_matches(svh_vo* v, const svh_p_match* matches, int32_t n) {
    svh::ActiveCaller active_;
    if (!v || (n > 0 && !matches) || n < 0) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    v->matched.assign(matches, matches + n);
    return update_motion(v);
}

int32_t svh_vo_estimate_motion(svh_vo* v, const svh_p_match* matches, int32_t n, double* tr_delta6) {
    svh::ActiveCaller active_;
    if (!v || !tr_delta6 || (n > 0 && !matches) || n < 0) return svh::fail(SVH_ERR_BAD_ARG, "null argument");
    return estimate(v, matches, n, tr_delta6);
}

void svh_vo_set_private_rand(svh_vo* v, int32_t enable, uint32_t seed) {
    if (!v) return;
    if (enable)
        v->rng.seed(seed);
    else
        v->rng.is_private = false;
}

void svh_rand_sequence(uint32_t seed, int32_t* out, int32_t n) {
    RandStream rs;
    rs.seed(seed);
    for (int32_t i = 0; i < n && out; i++) out[i] = rs.next();
}

void svh_vo_get_motion(svh_vo* v, double* Tr16) {
    if (v && Tr16) memcpy(Tr16, v->Tr, sizeof(v->Tr));
}

int32_t svh_vo_get_matches(svh_vo* v, svh_p_match* out, int32_t cap) {
    return v ? svh_matcher_get_matches(v->matcher, out, cap) : 0;
}

int32_t svh_vo_num_matches(svh_vo* v) { return v ? (int32_t)v->matched.size() : 0; }

int32_t svh_vo_get_inliers(svh_vo* v, int32_t* out, int32_t cap) {
    if (!v) return 0;
    for (int32_t i = 0; i < (int32_t)v->inliers.size() && i < cap && out; i++) out[i] = v->inliers[i];
    return (int32_t)v->inliers.size();
}

float svh_vo_get_gain(svh_vo* v, const int32_t* inliers, int32_t n) {
    return v ? svh_matcher_get_gain(v->matcher, inliers, n) : 1.f;
}

svh_matcher* svh_vo_matcher(svh_vo* v) { return v ? v->matcher : nullptr; }

}  // extern "C"
