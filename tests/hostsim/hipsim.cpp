// hipsim.cpp — TEST INFRASTRUCTURE ONLY (see hipsim.h).
#include "hipsim.h"

uint3_sim threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace hipsim {
static State g_state;
static unsigned long g_progress = 0;
State& st() { return g_state; }
void note_progress() { ++g_progress; }

static const size_t kStack = 256 * 1024;

#if HIPSIM_FAST_SWITCH
// void hipsim_switch(Ctx* from, Ctx* to): save the SysV callee-saved registers and the stack pointer of the running fiber,
// load those of the target, return into it
extern "C" void hipsim_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipsim_switch,.-hipsim_switch
)");
static inline void ctx_switch(Ctx* from, Ctx* to) { hipsim_switch(from, to); }
static void ctx_make(Ctx* c, char* stack, size_t size, void (*entry)()) {
  // entry must see rsp == 8 (mod 16), as right after a call: the return slot sits on a 16-byte boundary
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  void** sp = (void**)(top - 16);
  *sp = (void*)entry;  // popped by the final `ret` of hipsim_switch
  for (int i = 0; i < 6; ++i) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
  c->rsp = sp;
}
#else
static inline void ctx_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
static void ctx_make(Ctx* c, char* stack, size_t size, void (*entry)()) {
  getcontext(&c->uc);
  c->uc.uc_stack.ss_sp = stack;
  c->uc.uc_stack.ss_size = size;
  c->uc.uc_link = nullptr;
  makecontext(&c->uc, entry, 0);
}
#endif

void yield() {
  State& s = g_state;
  int me = s.cur;
  ctx_switch(&s.fibers[me].ctx, &s.sched);
  threadIdx = s.fibers[me].tid;
}

void wave_barrier() {
  State& s = g_state;
  int w = s.cur >> 6;
  int nl = s.nthreads - w * 64;
  if (nl > 64) nl = 64;
  unsigned g = s.wgen[w];
  if (++s.warrived[w] == nl) {
    s.warrived[w] = 0;
    s.wgen[w]++;
    ++g_progress;
  } else {
    while (s.wgen[w] == g) yield();
  }
}

static void trampoline() {
  State& s = g_state;
  s.body();
  s.fibers[s.cur].done = true;
  ctx_switch(&s.fibers[s.cur].ctx, &s.sched);
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = g_state;
  int n = (int)(block.x * block.y * block.z);
  if (n > 1024) { fprintf(stderr, "hipsim: block too large\n"); abort(); }
  if ((int)s.fibers.size() < n) {
    size_t old = s.fibers.size();
    s.fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)malloc(kStack);
  }
  s.body = body;
  s.nthreads = n;
  blockDim = block;
  gridDim = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = uint3_sim{bx, by, bz};
        s.arrived = 0;
        memset(s.warrived, 0, sizeof(s.warrived));
        for (int t = 0; t < n; ++t) {
          Fiber& f = s.fibers[t];
          f.done = false;
          f.xph = 0;
          f.tid = uint3_sim{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
          ctx_make(&f.ctx, f.stack, kStack, trampoline);
        }
        int remaining = n;
        while (remaining > 0) {
          unsigned long before = g_progress;
          for (int t = 0; t < n; ++t) {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = t;
            threadIdx = f.tid;
            ctx_switch(&s.sched, &f.ctx);
            if (f.done) { --remaining; ++g_progress; }
          }
          if (g_progress == before) { fprintf(stderr, "hipsim: deadlock (divergent barrier / shuffle?)\n"); abort(); }
        }
      }
  s.cur = -1;
}
}  // namespace hipsim
