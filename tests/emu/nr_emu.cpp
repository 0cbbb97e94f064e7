// Fiber scheduler for the wave-level emulator (see nr_prims.h in this directory).
#include "nr_prims.h"

namespace nr_emu {

BlockState* g_blk = nullptr;

// x86-64 SysV context switch: save callee-saved registers on the current stack,
// store sp, load the other sp, restore, ret.
asm(R"(
.text
.globl nr_emu_switch
.type nr_emu_switch,@function
nr_emu_switch:
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
.size nr_emu_switch,.-nr_emu_switch
)");

static void fiber_entry() {
  BlockState* b = g_blk;
  b->body();
  Fiber& f = b->fibers[b->cur];
  dma_land(0);
  f.done = true;
  b->alive--;
  // a thread that exits releases barriers the way a terminated wave does
  if (b->alive > 0 && b->arrived == b->alive) { b->arrived = 0; b->gen++; }
  nr_emu_switch(&f.sp, b->sched_sp);
  abort();
}

static const size_t kStack = 256 * 1024;

void launch(Dim3 grid, Dim3 block, size_t smem_bytes, std::function<void()> body) {
  BlockState blk;
  int nt = (int)(block.x * block.y * block.z);
  blk.nthreads = nt;
  blk.fibers.resize(nt);
  blk.waves.resize((nt + 63) / 64);
  blk.bdim = block;
  blk.gdim = grid;
  blk.body = body;
  const size_t guard = 4096;
  std::vector<unsigned char> smem(smem_bytes + guard);
  blk.smem = smem.data();
  blk.smem_bytes = smem_bytes;
  std::vector<unsigned char> stacks((size_t)nt * kStack);
  BlockState* prev = g_blk;
  g_blk = &blk;
  for (unsigned bz = 0; bz < grid.z; ++bz)
  for (unsigned by = 0; by < grid.y; ++by)
  for (unsigned bx = 0; bx < grid.x; ++bx) {
    blk.bid = Dim3{bx, by, bz};
    blk.alive = nt; blk.arrived = 0; blk.gen = 0;
    for (auto& w : blk.waves) { w.arrived = 0; w.gen = 0; }
    memset(smem.data(), 0xCB, smem_bytes);          // poison LDS: uninitialised reads are loud
    memset(smem.data() + smem_bytes, 0xA5, guard);
    for (int t = 0; t < nt; ++t) {
      Fiber& f = blk.fibers[t];
      f.done = false;
      f.dma.clear();
      f.tid = Dim3{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
      unsigned char* top = stacks.data() + (size_t)(t + 1) * kStack;
      uintptr_t sp = ((uintptr_t)top & ~(uintptr_t)15);
      uint64_t* s = (uint64_t*)sp;
      // layout expected by nr_emu_switch: r15 r14 r13 r12 rbx rbp ret
      *--s = 0;                        // fake return address slot alignment
      *--s = (uint64_t)&fiber_entry;   // ret target
      for (int i = 0; i < 6; ++i) *--s = 0;
      f.sp = s;
    }
    int remaining = nt;
    long spins = 0;
    while (remaining > 0) {
      int progressed = 0;
      for (int t = 0; t < nt; ++t) {
        Fiber& f = blk.fibers[t];
        if (f.done) continue;
        blk.cur = t;
        nr_emu_switch(&blk.sched_sp, f.sp);
        if (f.done) { remaining--; }
        progressed++;
      }
      if (++spins > 50000000L) { fprintf(stderr, "nr_emu: deadlock in block %u\n", bx); abort(); }
      (void)progressed;
    }
    for (size_t i = 0; i < guard; ++i)
      if (smem[smem_bytes + i] != 0xA5) { fprintf(stderr, "nr_emu: LDS overrun in block %u (+%zu)\n", bx, i); abort(); }
  }
  g_blk = prev;
}

}  // namespace nr_emu
