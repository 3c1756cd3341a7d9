// stream_search.h -- the lock-step search of a STREAM of samples on one device (vb2_cohort_run, round 4).
//
// Batch::optimize searches a fixed set of samples: the set's last steps serve the one or two slowest searches, and the
// next set starts only when those are over.  Here the device keeps `capacity` slots (two lanes taking turns, as in
// Batch::optimize): a sample that has converged hands its slot to the next sample that is ready -- the idea of continuous
// batching in an inference server -- so every step carries (nearly) a full load as long as samples keep coming, and nothing
// waits for a group to fill.  Every slot has the same number of 8-wave workgroups (cohort_waves(): 2 x num_cu / lane of them) whatever its neighbours are, so a
// sample's estimate does not depend on which samples it met on the device, nor on when it arrived: a run is reproducible
// bit for bit although its schedule is not.
#ifndef VB2_STREAM_SEARCH_H_
#define VB2_STREAM_SEARCH_H_

#include <hip/hip_runtime_api.h>

#include "../../include/vb2_abi.h"

namespace vb2 {

class Context;

struct StreamSource {
    static constexpr int kNone = -1, kEnd = -2;
    virtual ~StreamSource() {}
    // A sample that is ready to be searched: its id (>= 0) and context.  kNone: none is ready right now (only when block is
    // false); kEnd: none will come any more.  block: wait for one.
    virtual int next(bool block, Context** ctx) = 0;
    virtual const vb2_model& model(int id) = 0;
    // the search of sample id is over (rc != 0: failed, est undefined); its context is no longer used by the search
    virtual void done(int id, int rc, const vb2_estimate& est, double seconds) = 0;
};

// Runs until the source ends and every sample it delivered is done.  Non-zero: a device-level failure (every sample still
// in a slot has been reported done with that code).
// lane_streams: nullptr, or the two streams the lanes launch on (the caller's, outliving the call: cohort.cpp makes them
// apart from the streams its contexts are created on)
int stream_search(int device, int num_pc, int num_cu, int capacity, StreamSource& src, const hipStream_t* lane_streams = nullptr);
// what a reader thread prepares for a sample that will be searched by stream_search(capacity): its schedules
int prepare_for_stream(Context* c, int capacity);

}  // namespace vb2
#endif
