// xq_replay.hip — device-resident replay buffer fed by the sample records.
//
// Reference consumer (trainer.py:22-44, 309-321): a deque(maxlen=BUFFER_SIZE) of
// (board, move_probs, reward) samples; ReplayBuffer.push appends a game's samples in order,
// ReplayBuffer.sample draws np.random.choice(len, batch, replace=False) indices, and train_network
// re-encodes every sampled board on the CPU with the player plane hard-wired to red
// (encode_board(board, 1)) and uses only the reward as the value target.
//
// Here the samples never leave HBM: the engine's xq_sample_record[] (or the all-gathered shards)
// are compacted into a ring of records with the deque's drop-oldest semantics, and a batch is one
// gather + encode kernel that writes float32 states [B][15][10][9] and targets [B][1] in place.
#include "../../include/xq_selfplay.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace {

// records are [n_games][70]; the valid ones of game g are its first n_valid[g] plies (they were
// pushed in game order, ply order: the order trainer.py:224 / ReplayBuffer.push would append them)
__global__ void k_count_valid(const xq_sample_record *rec, int n_games, int32_t *n_valid)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_games) return;
    int n = 0;
    for (int i = 0; i < XQ_MAX_PLIES; i++) n += rec[(size_t)g * XQ_MAX_PLIES + i].valid ? 1 : 0;
    n_valid[g] = n;
}

// exclusive scan of n_valid (one workgroup; n_games <= 1024 * 1024)
__global__ __launch_bounds__(1024) void k_scan(const int32_t *n_valid, int n_games, int64_t *offset, int64_t *total)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n_games + 1023) / 1024;
    int64_t s = 0;
    for (int k = 0; k < per; k++) {
        const int g = t * per + k;
        if (g < n_games) s += n_valid[g];
    }
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = part[t] - s;
    for (int k = 0; k < per; k++) {
        const int g = t * per + k;
        if (g < n_games) { offset[g] = run; run += n_valid[g]; }
    }
    if (t == 1023) *total = part[1023];
}

// k-th new sample goes to ring slot (tail + k) % cap; only the last `cap` of them survive
__global__ void k_push(const xq_sample_record *rec, int n_games, const int64_t *offset, int64_t total,
                       xq_sample_record *ring, int64_t cap, int64_t tail)
{
    const int g = blockIdx.x;
    const size_t words = sizeof(xq_sample_record) / 16;                 // 36 x 16 B
    for (int i = 0; i < XQ_MAX_PLIES; i++) {
        const xq_sample_record *src = rec + (size_t)g * XQ_MAX_PLIES + i;
        if (!src->valid) break;
        const int64_t k = offset[g] + i;
        if (k < total - cap) continue;
        xq_sample_record *dst = ring + (size_t)((tail + k) % cap);
        for (size_t wd = threadIdx.x; wd < words; wd += blockDim.x)
            reinterpret_cast<uint4 *>(dst)[wd] = reinterpret_cast<const uint4 *>(src)[wd];
    }
}

// trainer.py:313-321: states = encode_board(board, 1) (neural_network.py:128-146, player plane all
// ones), target = float32(reward)
__global__ __launch_bounds__(64) void k_encode(const xq_sample_record *ring, int64_t cap, int64_t head,
                                               const int64_t *idx, int batch, float *states, float *targets)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const xq_sample_record *r = ring + (size_t)((head + idx[b]) % cap);
    __shared__ int8_t bd[96];
    if (lane < 12) {
        const uint32_t w = r->board[lane];
        for (int j = 0; j < 8; j++) {
            const uint32_t code = (w >> (4 * j)) & 15u;
            bd[8 * lane + j] = (int8_t)(code <= 7 ? (int)code : 7 - (int)code);
        }
    }
    __syncthreads();
    float *o = states + (size_t)b * 1350;
    for (int e = lane; e < 1350; e += 64) {
        const int c = e / 90, s = e % 90;
        float v;
        if (c == 14) v = 1.0f;
        else v = bd[s] == (c < 7 ? c + 1 : -(c - 6)) ? 1.0f : 0.0f;
        o[e] = v;
    }
    if (lane == 0) targets[b] = (float)r->z;
}

struct Replay {
    int device;
    int64_t cap, head = 0, count = 0;
    xq_sample_record *ring = nullptr;
    int32_t *n_valid = nullptr;
    int64_t *offset = nullptr, *total = nullptr, *idx = nullptr;
    int64_t n_games_cap = 0, idx_cap = 0;
};

thread_local std::string g_rerr;

}  // namespace

extern "C" const char *xq_replay_last_error(void) { return g_rerr.c_str(); }

#define RCHK(expr)                                                                           \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) { g_rerr = std::string(#expr) + ": " + hipGetErrorString(_e); return XQ_E_HIP; } \
    } while (0)

extern "C" int xq_replay_create(int device, int64_t capacity, void **out)
{
    if (!out || capacity <= 0) return XQ_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_rerr = "no HIP device visible"; return XQ_E_NOGPU; }
    RCHK(hipSetDevice(device));
    Replay *r = new Replay();
    r->device = device;
    r->cap = capacity;
    RCHK(hipMalloc(&r->ring, (size_t)capacity * sizeof(xq_sample_record)));
    RCHK(hipMalloc(&r->total, 8));
    *out = r;
    return 0;
}

extern "C" void xq_replay_destroy(void *h)
{
    Replay *r = reinterpret_cast<Replay *>(h);
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipFree(r->ring); (void)hipFree(r->total); (void)hipFree(r->n_valid); (void)hipFree(r->offset); (void)hipFree(r->idx);
    delete r;
}

extern "C" int64_t xq_replay_size(void *h) { return h ? reinterpret_cast<Replay *>(h)->count : 0; }

/* ReplayBuffer.push for every game of a record array [n_games][70] (device memory), in game order. */
extern "C" int xq_replay_push_records(void *h, void *stream, const void *records_dev, int n_games, int64_t *n_pushed)
{
    Replay *r = reinterpret_cast<Replay *>(h);
    if (!r || !records_dev || n_games <= 0) return XQ_E_INVALID;
    RCHK(hipSetDevice(r->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n_games > r->n_games_cap) {
        (void)hipFree(r->n_valid); (void)hipFree(r->offset);
        RCHK(hipMalloc(&r->n_valid, (size_t)n_games * 4));
        RCHK(hipMalloc(&r->offset, (size_t)n_games * 8));
        r->n_games_cap = n_games;
    }
    const xq_sample_record *rec = reinterpret_cast<const xq_sample_record *>(records_dev);
    hipLaunchKernelGGL(k_count_valid, dim3((n_games + 255) / 256), dim3(256), 0, s, rec, n_games, r->n_valid);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, r->n_valid, n_games, r->offset, r->total);
    int64_t total = 0;
    RCHK(hipMemcpyAsync(&total, r->total, 8, hipMemcpyDeviceToHost, s));
    RCHK(hipStreamSynchronize(s));
    const int64_t tail = (r->head + r->count) % r->cap;
    hipLaunchKernelGGL(k_push, dim3(n_games), dim3(64), 0, s, rec, n_games, r->offset, total, r->ring, r->cap, tail);
    RCHK(hipGetLastError());
    // deque(maxlen): the buffer now holds the last new_count items of (old items ++ new items)
    const int64_t new_count = r->count + total < r->cap ? r->count + total : r->cap;
    const int64_t end = (tail + total - 1 + r->cap) % r->cap;           // slot of the newest item
    if (total > 0) r->head = ((end - new_count + 1) % r->cap + r->cap) % r->cap;
    r->count = new_count;
    if (n_pushed) *n_pushed = total;
    return 0;
}

/* states float32 [batch][15][10][9], targets float32 [batch][1] for the logical (deque) indices. */
extern "C" int xq_replay_encode_batch(void *h, void *stream, const int64_t *idx_host, int batch, void *states_dev,
                                      void *targets_dev)
{
    Replay *r = reinterpret_cast<Replay *>(h);
    if (!r || !idx_host || batch <= 0 || !states_dev || !targets_dev) return XQ_E_INVALID;
    for (int i = 0; i < batch; i++)
        if (idx_host[i] < 0 || idx_host[i] >= r->count) { g_rerr = "index out of range"; return XQ_E_INVALID; }
    RCHK(hipSetDevice(r->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (batch > r->idx_cap) {
        (void)hipFree(r->idx);
        RCHK(hipMalloc(&r->idx, (size_t)batch * 8));
        r->idx_cap = batch;
    }
    RCHK(hipMemcpyAsync(r->idx, idx_host, (size_t)batch * 8, hipMemcpyHostToDevice, s));
    RCHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_encode, dim3(batch), dim3(64), 0, s, r->ring, r->cap, r->head, r->idx, batch,
                       reinterpret_cast<float *>(states_dev), reinterpret_cast<float *>(targets_dev));
    RCHK(hipGetLastError());
    return 0;
}

/* compatibility view: copy the records of the given logical indices to host memory */
extern "C" int xq_replay_read_records(void *h, void *stream, const int64_t *idx_host, int batch, void *records_host)
{
    Replay *r = reinterpret_cast<Replay *>(h);
    if (!r || !idx_host || batch <= 0 || !records_host) return XQ_E_INVALID;
    RCHK(hipSetDevice(r->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int i = 0; i < batch; i++) {
        if (idx_host[i] < 0 || idx_host[i] >= r->count) { g_rerr = "index out of range"; return XQ_E_INVALID; }
        const int64_t phys = (r->head + idx_host[i]) % r->cap;
        RCHK(hipMemcpyAsync(reinterpret_cast<xq_sample_record *>(records_host) + i, r->ring + phys, sizeof(xq_sample_record),
                            hipMemcpyDeviceToHost, s));
    }
    RCHK(hipStreamSynchronize(s));
    return 0;
}
