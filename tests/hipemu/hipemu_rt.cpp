// TEST INFRASTRUCTURE ONLY: runtime half of the host HIP emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

asm(R"(
.text
.globl zt_emu_switch
.type zt_emu_switch,@function
zt_emu_switch:
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
)");

namespace emu {

State& S() {
  static State s;
  return s;
}

void fiber_entry() {
  State& s = S();
  s.body();
  Fiber* f = s.cur;
  f->done = true;
  Block& b = s.blk;
  b.alive--;
  block_release_if_complete(b);
  Wave& w = b.waves[f->lin >> 6];
  w.alive--;
  wave_release_if_complete(w);
  zt_emu_switch(&f->sp, s.main_sp);
  abort();   // never resumed
}

static void prepare(Fiber& f, char* stack) {
  f.stack = stack;
  f.done = false;
  uintptr_t top = ((uintptr_t)stack + kStack) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                 // fake return address slot for fiber_entry
  *--sp = (void*)&fiber_entry;     // popped by `ret`
  for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
  f.sp = (void*)sp;
}

void launch(dim3 grid, dim3 block, std::function<void()> body) {
  State& s = S();
  int nthreads = block.x * block.y * block.z;
  if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "emu: bad block size %d\n", nthreads); abort(); }
  while ((int)s.stacks.size() < nthreads) s.stacks.push_back((char*)aligned_alloc(64, kStack));
  s.fibers.resize(nthreads);
  s.body = body;
  s.bdim = {block.x, block.y, block.z};
  s.gdim = {grid.x, grid.y, grid.z};
  int nwaves = (nthreads + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.bid = {bx, by, bz};
        Block& b = s.blk;
        b.nthreads = b.alive = nthreads;
        b.arrived = 0;
        b.gen = 0;
        b.waves.assign(nwaves, Wave());
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = s.fibers[t];
          prepare(f, s.stacks[t]);
          f.lin = t;
          f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
          b.waves[t >> 6].alive++;
        }
        int remaining = nthreads;
        long spins = 0;
        while (remaining > 0) {
          int progressed = 0;
          for (int t = 0; t < nthreads; ++t) {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = &f;
            zt_emu_switch(&s.main_sp, f.sp);
            if (f.done) { remaining--; progressed++; }
          }
          if (++spins > 50000000L) { fprintf(stderr, "emu: deadlock suspected (barrier mismatch)\n"); abort(); }
        }
      }
  s.cur = nullptr;
}

}  // namespace emu
