// Batched launches for K Matcher / visual-odometry objects driven in lockstep (svh_matcher_*_batch,
// svh_vo_process_batch): while a recorder is installed on the calling thread, the kernel launchers
// (mlaunch_*, vlaunch_*) do not launch -- they append their arguments as one JOB to the recorder's slot for
// that call position.  Every object issues the same sequence of launcher calls (same parameters and image
// size: the batch entries check it), so slot c holds the K jobs of the c-th call, and flush() turns each slot
// into ONE launch of the kernel's batched form (blockIdx.z = job, grid = the largest job's).  A frame of K
// sequences then costs ~25 launches instead of ~45 K.
#ifndef SVH_BATCH_REC_H
#define SVH_BATCH_REC_H

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <vector>

namespace svh {

// launches the batched form of one kernel: jobs = device copy of the job table
typedef void (*BatchLaunchFn)(const void* jobs, int njobs, unsigned gx, unsigned gy, size_t lds, hipStream_t s);

struct BatchRec {
    struct Slot {
        BatchLaunchFn fn = nullptr;
        unsigned gx = 0, gy = 1;
        size_t lds = 0, job_bytes = 0;
        int njobs = 0;
        std::vector<uint8_t> jobs;
    };
    std::vector<Slot> slots;
    int cursor = 0;           // call position of the object being recorded
    bool broken = false;      // an object issued a different call sequence than the first one
    // side streams (+ one event each) for work a lockstep call issues outside the recorded sequence: the image
    // uploads, several in flight at once
    static constexpr int kSide = 3;
    hipStream_t side[kSide] = {nullptr, nullptr, nullptr};
    hipEvent_t side_done[kSide] = {nullptr, nullptr, nullptr};
    // a recorder whose launches outlive the call that flushed them (prefetch): flush() notes its stream, reuse()
    // waits for that stream before the arena is written again
    bool track = false, flush_pending = false;
    hipStream_t last_stream = nullptr;
    hipError_t reuse();
    hipError_t ensure_side();                 // creates them on first use (current device)
    hipError_t join_side(hipStream_t s);      // s waits for everything issued on the side streams so far
    uint8_t* h_arena = nullptr;   // pinned staging of the job tables
    uint8_t* d_arena = nullptr;
    size_t cap = 0, used = 0;     // `used` advances per flush, reset by synced()

    void begin_object() { cursor = 0; }
    void reset() {   // forget recorded jobs (a new phase, or after a mismatch)
        slots.clear();
        cursor = 0;
        broken = false;
    }
    template <class J>
    void add(BatchLaunchFn fn, const J& job, unsigned gx, unsigned gy = 1, size_t lds = 0) {
        if ((size_t)cursor == slots.size()) {
            slots.emplace_back();
            slots.back().fn = fn;
            slots.back().job_bytes = sizeof(J);
        }
        Slot& s = slots[cursor++];
        if (s.fn != fn || s.job_bytes != sizeof(J)) {
            broken = true;
            return;
        }
        s.gx = gx > s.gx ? gx : s.gx;
        s.gy = gy > s.gy ? gy : s.gy;
        s.lds = lds > s.lds ? lds : s.lds;
        const size_t at = s.jobs.size();
        s.jobs.resize(at + sizeof(J));
        memcpy(s.jobs.data() + at, &job, sizeof(J));
        s.njobs++;
    }
    // job tables -> device (one copy), one launch per slot, slots cleared.  hipSuccess or the first error
    hipError_t flush(hipStream_t s);
    void synced() { used = 0; }   // the stream was waited for: the arena may be reused from its start
    void release();   // frees the arena and the side streams
    ~BatchRec() { release(); }
    BatchRec() = default;
    BatchRec(const BatchRec&) = delete;
    BatchRec& operator=(const BatchRec&) = delete;
};

// non-null: the launchers record into it (set and cleared by the batch entries on their own thread)
extern thread_local BatchRec* t_rec;
// the calling thread's recorder for a device (arena and side streams are kept for the thread's lifetime)
BatchRec& batch_recorder(int device);
BatchRec& prefetch_recorder(int device);
// fn(0..n-1) on the library's parked helper threads and the caller; returns when all are done
void batch_parallel_for(int n, const std::function<void(int)>& fn);

}  // namespace svh
#endif
