// hipemu runtime: fiber scheduler for tests/hipemu/hip/hip_runtime.h (test infrastructure only).
#include <hip/hip_runtime.h>

#include <sys/mman.h>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

static constexpr size_t STACK_BYTES = 256 * 1024;

Ctx& ctx() {
    static Ctx c;
    return c;
}

static void fiber_exit() {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    f->finished = true;
    ++c.progress;
    Sync& w = c.waves[f->wave];
    ++w.finished;
    if (w.arrived > 0 && w.arrived + w.finished == w.total) w.complete();
    Sync& b = c.block;
    ++b.finished;
    if (b.arrived > 0 && b.arrived + b.finished == b.total) b.complete();
    void* dummy;
    hipemu_switch(&dummy, c.sched_sp);
    std::abort();
}

static void trampoline() {
    ctx().body();
    fiber_exit();
}

void yield_wait(Sync* s, uint64_t g) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    f->wait = s;
    f->wait_gen = g;
    hipemu_switch(&f->sp, c.sched_sp);
}

static void prepare(Fiber& f) {
    if (!f.stack) {
        void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("hipemu mmap"); std::abort(); }
        f.stack = (char*)p;
    }
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // fake return address of trampoline (keeps rsp = 8 mod 16 at entry)
    *--sp = (void*)&trampoline;      // popped by `ret` in hipemu_switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
    f.sp = sp;
}

void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    Ctx& c = ctx();
    const int n = (int)(block.x * block.y * block.z);
    if (n <= 0 || n > 1024 || grid.x == 0 || grid.y == 0 || grid.z == 0) { c.last_error = 1; return; }
    const int nw = (n + WAVE - 1) / WAVE;
    if ((int)c.fibers.size() < n) c.fibers.resize(n);
    if ((int)c.waves.size() < nw) c.waves.resize(nw);
    if (c.dyn_lds.size() < shmem + 64) c.dyn_lds.resize(shmem + 64);
    c.bdim = block; c.gdim = grid; c.body = std::move(body);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                c.bid = {bx, by, bz};
                for (int t = 0; t < n; ++t) {
                    Fiber& f = c.fibers[t];
                    f.finished = false; f.wait = nullptr; f.flat = t; f.lane = t % WAVE; f.wave = t / WAVE;
                    f.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
                    prepare(f);
                }
                for (int w = 0; w < nw; ++w) c.waves[w].init(std::min(WAVE, n - w * WAVE), WAVE_PAY);
                c.block.init(n, 4);
                int remaining = n;
                while (remaining > 0) {
                    bool ran = false;
                    for (int t = 0; t < n; ++t) {
                        Fiber& f = c.fibers[t];
                        if (f.finished) continue;
                        if (f.wait) { if (f.wait->gen == f.wait_gen) continue; f.wait = nullptr; }
                        c.cur = &f; ran = true;
                        hipemu_switch(&c.sched_sp, f.sp);
                        if (f.finished) --remaining;
                    }
                    if (!ran) {
                        fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): %d threads blocked on a collective that can "
                                        "never complete (divergent __syncthreads / wave op?)\n", bx, by, bz, remaining);
                        for (int t = 0; t < n && t < 1024; ++t)
                            if (!c.fibers[t].finished) { fprintf(stderr, "  first blocked thread %d (%s sync)\n", t, c.fibers[t].wait == &c.block ? "block" : "wave"); break; }
                        std::abort();
                    }
                }
            }
    c.cur = nullptr;
}

}  // namespace hipemu
