// lockstep.cpp -- see lockstep.h.
#include "lockstep.h"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>

namespace vb2 {

namespace {
constexpr size_t kFiberStack = 256 * 1024;      // the optimiser's frames are a few KB deep
size_t page_size()
{
    static const size_t p = (size_t)std::max(4096L, sysconf(_SC_PAGESIZE));
    return p;
}
}

FiberGang::FiberGang(int num_fiber, int max_points_per_request)
    : fibers_(num_fiber), cb_(num_fiber), req_(num_fiber), max_points_(std::max(1, max_points_per_request))
{
    for (int i = 0; i < num_fiber; ++i) cb_[i] = Cb{this, i};
}

FiberGang::~FiberGang()
{
    for (Fiber& f : fibers_)
        if (f.map) munmap(f.map, kFiberStack + page_size());
}

void FiberGang::entry(unsigned lo, unsigned hi)
{
    FiberGang* g = reinterpret_cast<FiberGang*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    const int i = g->cur_;
    g->body_(i);                     // (never throws: the body catches)
    g->fibers_[i].done = true;
    g->req_[i].n = 0;
}                                    // uc_link: back to run()

int FiberGang::eval_cb(void* user, int32_t n, const double* p1, const double* p2, const double* a, double* o)
{
    Cb* cb = static_cast<Cb*>(user);
    FiberGang* g = cb->gang;
    const int k = g->num_pc_;
    for (int done = 0; done < n; done += g->max_points_) {
        Request& r = g->req_[cb->index];
        r.n = std::min(g->max_points_, n - done);
        r.p1 = p1 + (size_t)done * k;
        r.p2 = p2 + (size_t)done * k;
        r.a = a + done;
        r.out = o + done;
        swapcontext(&g->fibers_[cb->index].ctx, &g->main_);      // parked until the step has been evaluated
        if (g->error_) return g->error_;
    }
    return 0;
}

int FiberGang::prepare(int i)
{
    Fiber& f = fibers_[i];
    if (!f.map) {
        // stacks grow downwards: an overflow hits the PROT_NONE page and faults instead of
        // scribbling over the heap
        void* m = mmap(nullptr, kFiberStack + page_size(), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return -1;
        (void)mprotect(m, page_size(), PROT_NONE);
        f.map = m;
    }
    f.done = false;
    req_[i] = Request{};
    if (getcontext(&f.ctx) != 0) return -1;
    f.ctx.uc_stack.ss_sp = static_cast<char*>(f.map) + page_size();
    f.ctx.uc_stack.ss_size = kFiberStack;
    f.ctx.uc_link = &main_;
    const uintptr_t p = reinterpret_cast<uintptr_t>(this);
    makecontext(&f.ctx, reinterpret_cast<void (*)()>(entry), 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    return 0;
}

int FiberGang::start(int num_pc, const std::function<void(int)>& body)
{
    num_pc_ = num_pc;
    body_ = body;
    error_ = 0;
    const int n = (int)fibers_.size();
    for (int i = 0; i < n; ++i)
        if (prepare(i) < 0) return -1;
    for (int i = 0; i < n; ++i) {                                  // up to everybody's first request
        cur_ = i;
        swapcontext(&main_, &fibers_[i].ctx);
    }
    return 0;
}

void FiberGang::open(int num_pc, const std::function<void(int)>& body)
{
    num_pc_ = num_pc;
    body_ = body;
    error_ = 0;
    for (size_t i = 0; i < fibers_.size(); ++i) {
        fibers_[i].done = true;
        req_[i] = Request{};
    }
}

int FiberGang::spawn(int i)
{
    if (prepare(i) < 0) {
        fibers_[i].done = true;
        return -1;
    }
    cur_ = i;
    swapcontext(&main_, &fibers_[i].ctx);                          // up to its first request, or to the end
    return 0;
}

bool FiberGang::pending() const
{
    for (size_t i = 0; i < fibers_.size(); ++i)
        if (!fibers_[i].done && req_[i].n > 0) return true;
    return false;
}

void FiberGang::resume_parked()
{
    const int n = (int)fibers_.size();
    for (int i = 0; i < n; ++i) {
        if (fibers_[i].done || req_[i].n <= 0) continue;
        req_[i].n = 0;
        cur_ = i;
        swapcontext(&main_, &fibers_[i].ctx);                      // up to its next request, or to the end
    }
}

int FiberGang::run(int num_pc, const std::function<void(int)>& body, const StepFn& step)
{
    if (start(num_pc, body) < 0) return -1;
    while (pending()) {
        if (!error_) {
            ++steps;
            const int rc = step(req_);
            if (rc) error_ = rc;                                   // the fibers see it and unwind
        }
        resume_parked();
    }
    return error_;
}

}  // namespace vb2
