// lockstep.h -- several optimiser runs advancing in lock-step on ONE host thread.
//
// Every run (a sample of a cohort, a restart of a multi-start search) executes ordinary blocking
// code -- Estimator::OptimizeLLK with the reference-exact simplex -- as a FIBER: its own small
// stack, switched with swapcontext.  A fiber runs until its optimiser asks for likelihood values,
// parks the request and yields; when every live fiber has parked, the step callback evaluates all
// requests together (one kernel launch), the values are handed back and the fibers run on.  No
// threads, no locks, no wake-ups: a step costs the host ~0.3 us per run where one thread per run
// cost a futex round trip each (9 ms of CPU per sample and search at C3 size, and under a
// container's CPU quota it throttled the readers and the search alike).
#ifndef VB2_LOCKSTEP_H_
#define VB2_LOCKSTEP_H_

#include <ucontext.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

namespace vb2 {

class FiberGang {
public:
    // A parked request: n points (rows of pc1 / pc2 with num_pc columns, alpha) and where the
    // values go.  The pointers are into the fiber's own frames: alive while it is parked.
    struct Request {
        int n = 0;
        const double *p1 = nullptr, *p2 = nullptr, *a = nullptr;
        double* out = nullptr;
    };
    // evaluates every request with n > 0 (writes req[i].out[0..n)); non-zero = error for all
    typedef std::function<int(std::vector<Request>& req)> StepFn;

    explicit FiberGang(int num_fiber, int max_points_per_request);
    ~FiberGang();
    // the evaluator to hand to run i's Estimator (vb2_eval_fn signature) and its user pointer
    static int eval_cb(void* user, int32_t n, const double* p1, const double* p2, const double* a, double* o);
    void* user(int i) { return &cb_[i]; }
    // body(i) runs as fiber i; it must not let an exception escape
    int run(int num_pc, const std::function<void(int)>& body, const StepFn& step);
    int64_t steps = 0;

    // The same in pieces, for a caller that keeps several gangs going (one gang's step on the
    // device while another gang's fibers run on the host):
    int start(int num_pc, const std::function<void(int)>& body);   // every fiber up to its first request; < 0: no contexts
    bool pending() const;                                           // some fiber is parked with a request
    std::vector<Request>& requests() { return req_; }
    void fail(int rc) { if (!error_) error_ = rc; }                 // the fibers see it at resume and unwind
    int error() const { return error_; }
    void resume_parked();                                           // values delivered: every parked fiber runs on
    // A gang whose fibers come and go (the streaming cohort search: a finished sample's fiber is given the next sample):
    // open() starts with every fiber idle; spawn(i) runs body(i) as fiber i up to its first request (or to its end);
    // idle(i): fiber i is not running anything (never spawned, or its body has returned).
    void open(int num_pc, const std::function<void(int)>& body);
    int spawn(int i);                                               // < 0: no context / stack
    bool idle(int i) const { return fibers_[i].done; }
    int size() const { return (int)fibers_.size(); }

private:
    struct Fiber {
        ucontext_t ctx;
        void* map = nullptr;                // mmap'ed: one inaccessible guard page below the stack
        bool done = false;
    };
    struct Cb {
        FiberGang* gang;
        int index;
    };
    static void entry(unsigned lo, unsigned hi);
    ucontext_t main_;
    std::vector<Fiber> fibers_;
    std::vector<Cb> cb_;
    std::vector<Request> req_;
    std::function<void(int)> body_;
    int cur_ = -1, num_pc_ = 0, max_points_, error_ = 0;
    int prepare(int i);                  // fiber i's stack and context, ready to enter body(i)
};

}  // namespace vb2
#endif
