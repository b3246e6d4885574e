// simt.cpp — fiber scheduler of the wave64 SIMT emulator (see simt.h).  TEST INFRASTRUCTURE ONLY.
#include "simt.h"

#include <memory>
#include <mutex>

namespace simt {
thread_local Block* g_block = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

static constexpr size_t kStack = 512 * 1024;

static void fiber_entry() {
    Block* b = g_block;
    b->body();
    b->fibers[b->cur].done = true;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

static void run_block(Block& blk, std::vector<std::unique_ptr<char[]>>& stacks, dim3 block, unsigned bid, dim3 grid,
                      const std::function<void()>& body) {
    const unsigned nt = block.x * block.y * block.z;
    blk.nthreads = nt;
    blk.body = body;
    blk.fibers.resize(nt);
    blk.waves.assign((nt + WAVE - 1) / WAVE, WaveShared());
    blk.bar_arrived = blk.bar_gen = 0;
    while (stacks.size() < nt) stacks.emplace_back(new char[kStack]);
    g_block = &blk;
    t_blockDim = block;
    t_gridDim = grid;
    t_blockIdx = dim3(bid % grid.x, (bid / grid.x) % grid.y, bid / (grid.x * grid.y));
    for (unsigned t = 0; t < nt; ++t) {
        Fiber& f = blk.fibers[t];
        f.done = false;
        f.tid = t;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = stacks[t].get();
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    unsigned remaining = nt;
    unsigned long long idle_rounds = 0;
    while (remaining) {
        unsigned before_gen = blk.bar_gen, before_rem = remaining;
        unsigned long long wave_gens = 0;
        for (auto& w : blk.waves) wave_gens += w.gen;
        for (unsigned t = 0; t < nt; ++t) {
            Fiber& f = blk.fibers[t];
            if (f.done) continue;
            blk.cur = t;
            t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            swapcontext(&blk.sched, &f.ctx);
            if (f.done) --remaining;
        }
        unsigned long long wave_gens2 = 0;
        for (auto& w : blk.waves) wave_gens2 += w.gen;
        if (before_gen == blk.bar_gen && before_rem == remaining && wave_gens == wave_gens2) {
            if (++idle_rounds > 4) {
                fprintf(stderr, "simt: deadlock in block %u (a lane exited or diverged before a barrier / cross-lane op)\n", bid);
                abort();
            }
        } else {
            idle_rounds = 0;
        }
    }
    g_block = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const unsigned nblocks = grid.x * grid.y * grid.z;
    unsigned nthr = std::thread::hardware_concurrency();
    if (const char* e = getenv("KPN_SIMT_THREADS")) nthr = (unsigned)atoi(e);
    if (nthr < 1) nthr = 1;
    if (nthr > nblocks) nthr = nblocks;
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        Block blk;
        std::vector<std::unique_ptr<char[]>> stacks;
        for (;;) {
            unsigned b = next.fetch_add(1);
            if (b >= nblocks) break;
            run_block(blk, stacks, block, b, grid, body);
        }
    };
    if (nthr == 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthr; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
}
}  // namespace simt
