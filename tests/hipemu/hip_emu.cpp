// tests/hipemu/hip_emu.cpp -- TEST-ONLY fiber runtime behind tests/hipemu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <ucontext.h>

#include <vector>

// Dynamic LDS of the running launch: allocated per launch at EXACTLY the size the launch asked for (so that an index past
// the kernel's allocation is a heap overflow AddressSanitizer reports) and filled with 0xFF before every block (fp32 / f16
// NaN patterns, -1 as an integer: a kernel that consumes LDS it never wrote, e.g. stale data of the previous workgroup,
// poisons its result instead of passing by accident).
char* lm_emu_dynsmem = nullptr;
static size_t g_dynsmem_bytes = 0;

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif
#ifndef HIPEMU_ASAN
#define HIPEMU_ASAN 0
#endif

namespace hipemu {

enum { READY = 0, AT_BLOCK_BARRIER = 1, AT_WAVE_COLL = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = READY;
    int linear = 0;       // linear thread id in block
    dim3 tid;
    unsigned long long wait_gen = 0;
};

struct Wave {
    unsigned long long vals[2][64];
    unsigned long long mask[2];
    int arrived = 0;
    unsigned long long gen = 0;   // completed collectives
};

static const size_t kStack = 256 * 1024;
Fiber* g_cur = nullptr;
ThreadCtx g_ctx;

static ucontext_t g_sched;
static std::vector<Fiber> g_fibers;
static std::vector<Wave> g_waves;
static const std::function<void()>* g_body = nullptr;
static int g_block_arrived = 0;
static unsigned long long g_block_gen = 0;
static int g_live = 0;

#if HIPEMU_ASAN
// AddressSanitizer must be told about every stack switch (ucontext fibers), or it reports false stack overflows
static const void* g_sched_stack = nullptr;
static size_t g_sched_stack_size = 0;
static void to_sched(bool dying)
{
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(dying ? nullptr : &fake, g_sched_stack, g_sched_stack_size);
    swapcontext(&g_cur->ctx, &g_sched);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
}
static void to_fiber(Fiber& f)
{
    void* fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, f.stack, kStack);
    swapcontext(&g_sched, &f.ctx);
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
}
static void fiber_main()
{
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack, &g_sched_stack_size);
    (*g_body)();
    g_cur->state = DONE;
    to_sched(true);
}
static void yield_to_sched() { to_sched(false); }
#else
static void fiber_main()
{
    (*g_body)();
    g_cur->state = DONE;
    swapcontext(&g_cur->ctx, &g_sched);
}

static void yield_to_sched() { swapcontext(&g_cur->ctx, &g_sched); }
static void to_fiber(Fiber& f) { swapcontext(&g_sched, &f.ctx); }
#endif

int lane_id() { return g_cur->linear & 63; }

static int live_in_wave(int w, int nthreads)
{
    int n = 0;
    for (int l = 0; l < 64; l++) {
        int t = w * 64 + l;
        if (t < nthreads && g_fibers[t].state != DONE) n++;
    }
    return n;
}

void syncthreads()
{
    Fiber* f = g_cur;
    f->state = AT_BLOCK_BARRIER;
    f->wait_gen = g_block_gen;
    g_block_arrived++;
    yield_to_sched();
}

const unsigned long long* wave_gather(unsigned long long v, unsigned long long* active_mask)
{
    Fiber* f = g_cur;
    int w = f->linear >> 6, l = f->linear & 63;
    Wave& wv = g_waves[w];
    int slot = (int)(wv.gen & 1);
    if (wv.arrived == 0) {
        wv.mask[slot] = 0;
        for (int i = 0; i < 64; i++) wv.vals[slot][i] = 0;
    }
    wv.vals[slot][l] = v;
    wv.mask[slot] |= 1ull << l;
    wv.arrived++;
    f->state = AT_WAVE_COLL;
    f->wait_gen = wv.gen;
    yield_to_sched();
    // released: generation advanced; our data sits in `slot`
    if (active_mask) *active_mask = wv.mask[slot];
    return wv.vals[slot];
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem)
{
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || grid.x * grid.y * grid.z == 0) return;
    if (shmem > 160 * 1024) { fprintf(stderr, "hipemu: launch asks for %zu bytes of dynamic LDS (> 160 KB)\n", shmem); abort(); }
    char* const outer_smem = lm_emu_dynsmem;          // launches do not nest, but keep the state tidy
    const size_t outer_bytes = g_dynsmem_bytes;
    lm_emu_dynsmem = nullptr;
    if (shmem && posix_memalign((void**)&lm_emu_dynsmem, 64, shmem) != 0) abort();
    g_dynsmem_bytes = shmem;
    g_body = &body;
    if ((int)g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); i++) g_fibers[i].stack = (char*)malloc(kStack);
    }
    g_waves.assign((nthreads + 63) / 64, Wave());
    ThreadCtx saved = g_ctx;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        for (auto& w : g_waves) { w.arrived = 0; w.gen = 0; }
        if (shmem) memset(lm_emu_dynsmem, 0xff, shmem);
        g_block_arrived = 0;
        g_block_gen = 0;
        g_live = nthreads;
        for (int t = 0; t < nthreads; t++) {
            Fiber& f = g_fibers[t];
            f.state = READY;
            f.linear = t;
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = &g_sched;
            makecontext(&f.ctx, (void (*)())fiber_main, 0);
        }
        int idle_rounds = 0;
        while (g_live > 0) {
            bool progressed = false;
            for (int t = 0; t < nthreads; t++) {
                Fiber& f = g_fibers[t];
                if (f.state == DONE) continue;
                if (f.state == AT_BLOCK_BARRIER) {
                    if (f.wait_gen == g_block_gen) {
                        if (g_block_arrived < g_live) continue;
                        g_block_gen++;      // everyone alive has arrived: release this generation
                        g_block_arrived = 0;
                    }
                    f.state = READY;
                } else if (f.state == AT_WAVE_COLL) {
                    Wave& wv = g_waves[t >> 6];
                    if (f.wait_gen == wv.gen) {
                        if (wv.arrived < live_in_wave(t >> 6, nthreads)) continue;
                        wv.gen++;
                        wv.arrived = 0;
                    }
                    f.state = READY;
                }
                g_cur = &f;
                g_ctx.tid = f.tid;
                g_ctx.bid = dim3(bx, by, bz);
                g_ctx.bdim = block;
                g_ctx.gdim = grid;
                to_fiber(f);
                progressed = true;
                if (f.state == DONE) g_live--;
            }
            if (!progressed) {
                if (++idle_rounds > 2) {
                    fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): divergent barrier/collective\n", bx, by, bz);
                    abort();
                }
            } else {
                idle_rounds = 0;
            }
        }
    }
    g_ctx = saved;
    g_cur = nullptr;
    free(lm_emu_dynsmem);
    lm_emu_dynsmem = outer_smem;
    g_dynsmem_bytes = outer_bytes;
}

}  // namespace hipemu

// k-ordered fmaf chain, one rounding per product (MI355X_MICROARCH: exact f32, bitwise == fmaf chain)
lm_f32x16 hipemu_mfma_32x32x2f32(float a, float b, lm_f32x16 c)
{
    // A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    unsigned long long act;
    unsigned ai, bi;
    memcpy(&ai, &a, 4);
    memcpy(&bi, &b, 4);
    const unsigned long long* v = hipemu::wave_gather(((unsigned long long)bi << 32) | ai, &act);
    int lane = hipemu::lane_id();
    int col = lane & 31;
    for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; k++) {
            unsigned au = (unsigned)(v[row + 32 * k] & 0xffffffffu);
            unsigned bu = (unsigned)(v[col + 32 * k] >> 32);
            float af, bf;
            memcpy(&af, &au, 4);
            memcpy(&bf, &bu, 4);
            acc = fmaf(af, bf, acc);
        }
        c[r] = acc;
    }
    return c;
}

lm_f32x4 hipemu_mfma_16x16x4f32(float a, float b, lm_f32x4 c)
{
    // A[l&15][k = l>>4], B[k = l>>4][l&15]; C/D: col = lane&15, row = (lane>>4)*4 + reg
    unsigned long long act;
    unsigned ai, bi;
    memcpy(&ai, &a, 4);
    memcpy(&bi, &b, 4);
    const unsigned long long* v = hipemu::wave_gather(((unsigned long long)bi << 32) | ai, &act);
    int lane = hipemu::lane_id();
    int col = lane & 15;
    for (int r = 0; r < 4; r++) {
        int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; k++) {
            unsigned au = (unsigned)(v[row + 16 * k] & 0xffffffffu);
            unsigned bu = (unsigned)(v[col + 16 * k] >> 32);
            float af, bf;
            memcpy(&af, &au, 4);
            memcpy(&bf, &bu, 4);
            acc = fmaf(af, bf, acc);
        }
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_f16: lane l holds A[row l&31][k = 8*(l>>5) + j] and B[k = 8*(l>>5) + j][col l&31], j = 0..7
// (cdna_hip_programming.md section 3, bf16 map; the f16 form uses the same one); products exact in fp32, fp32 accumulate.
typedef _Float16 emu_h8 __attribute__((ext_vector_type(8)));
lm_f32x16 hipemu_mfma_32x32x16f16(emu_h8 a, emu_h8 b, lm_f32x16 c)
{
    static thread_local _Float16 A[64][8], B[64][8];
    unsigned long long raw[4];
    memcpy(raw, &a, 16);
    memcpy(raw + 2, &b, 16);
    unsigned long long act;
    const int lane = hipemu::lane_id();
    // four 8-byte gathers (the rendezvous returns every lane's value)
    unsigned long long all[4][64];
    for (int q = 0; q < 4; q++) {
        const unsigned long long* v = hipemu::wave_gather(raw[q], &act);
        memcpy(all[q], v, sizeof(all[q]));
    }
    for (int l = 0; l < 64; l++) {
        memcpy(&A[l][0], &all[0][l], 8); memcpy(&A[l][4], &all[1][l], 8);
        memcpy(&B[l][0], &all[2][l], 8); memcpy(&B[l][4], &all[3][l], 8);
    }
    const int col = lane & 31;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; h++)
            for (int j = 0; j < 8; j++) acc += (float)A[row + 32 * h][j] * (float)B[col + 32 * h][j];
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x32_f16: lane l holds A[row l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col l&15], j = 0..7;
// C/D: col = lane & 15, row = (lane >> 4) * 4 + reg (cdna_hip_programming.md section 3)
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
emu_f32x4 hipemu_mfma_16x16x32f16(emu_h8 a, emu_h8 b, emu_f32x4 c)
{
    static thread_local _Float16 A[64][8], B[64][8];
    unsigned long long raw[4];
    memcpy(raw, &a, 16);
    memcpy(raw + 2, &b, 16);
    unsigned long long act;
    const int lane = hipemu::lane_id();
    unsigned long long all[4][64];
    for (int q = 0; q < 4; q++) {
        const unsigned long long* v = hipemu::wave_gather(raw[q], &act);
        memcpy(all[q], v, sizeof(all[q]));
    }
    for (int l = 0; l < 64; l++) {
        memcpy(&A[l][0], &all[0][l], 8); memcpy(&A[l][4], &all[1][l], 8);
        memcpy(&B[l][0], &all[2][l], 8); memcpy(&B[l][4], &all[3][l], 8);
    }
    const int col = lane & 15;
    for (int r = 0; r < 4; r++) {
        const int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; g++)
            for (int j = 0; j < 8; j++) acc += (float)A[row + 16 * g][j] * (float)B[col + 16 * g][j];
        c[r] = acc;
    }
    return c;
}
