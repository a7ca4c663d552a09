// exchange.hip -- the multi-GPU side of the path, in C behind the ABI: range sharding by whole tiles and the ONE
// collective of a frame, the RCCL allgather of the compacted visible set (as its 1-bit-per-entity mask), followed by
// the local expansion into the identical ascending global id list on every rank.
//
// One process per GPU.  RCCL is not linked: the library the process already has (torch ships one) or the system's is
// opened at run time, so libclapgpu.so loads on machines without it and single-GPU users never touch it.  The unique
// id travels by whatever channel the launcher has (bench.py: a torch.distributed broadcast; an engine: its own
// socket) -- clapgpu_exchange_unique_id() on rank 0, the 128 bytes to everyone, clapgpu_exchange_create() everywhere.
// xGMI is point to point: at 8 GPUs a rank receives 7 x 125 KB per million entities -- latency-, not bandwidth-bound,
// which is why the payload is the mask and not the id list (~9x the bytes at 30 % visibility).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "common.h"

namespace {

typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_get_uid)(nccl_uid *);
typedef int (*fn_init_rank)(void **, int, nccl_uid, int);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
typedef int (*fn_comm_int)(void *, int *);

struct Rccl {
    void *lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    fn_comm_int comm_count = nullptr, comm_user_rank = nullptr;
};
Rccl g_rccl;
std::string g_rccl_path;

int load_rccl()
{
    if (g_rccl.lib) return CLAPGPU_OK;
    const char *cands[4] = { g_rccl_path.empty() ? nullptr : g_rccl_path.c_str(), getenv("CLAPGPU_RCCL_LIBRARY"), "librccl.so.1", "librccl.so" };
    void *h = nullptr;
    for (int pass = 0; pass < 2 && !h; pass++)                   // first a copy the process has already loaded
        for (const char *c : cands)
            if (c && !h) h = dlopen(c, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!h) return CLAPGPU_ERR_NOT_SUPPORTED;
    g_rccl.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    g_rccl.comm_count = (fn_comm_int)dlsym(h, "ncclCommCount");
    g_rccl.comm_user_rank = (fn_comm_int)dlsym(h, "ncclCommUserRank");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.allgather || !g_rccl.destroy) return CLAPGPU_ERR_NOT_SUPPORTED;
    g_rccl.lib = h;
    return CLAPGPU_OK;
}

constexpr int NCCL_UINT64 = 5;           // ncclUint64 (nccl.h)

} // namespace

struct clapgpu_exchange {
    void *comm;
    int rank, world;
};

extern "C" void clapgpu_exchange_set_library(const char *path) { g_rccl_path = path ? path : ""; }

// Can this process open RCCL at all?  Cheap (a dlopen + five dlsym, once), no communicator: every rank asks BEFORE any of
// them enters the collective ncclCommInitRank, so that a rank which cannot load the library does not leave the others
// waiting inside it (the launcher reduces the answers with MIN and only then calls clapgpu_exchange_create everywhere).
extern "C" int clapgpu_exchange_available(void) { return load_rccl() == CLAPGPU_OK ? 1 : 0; }

extern "C" int clapgpu_exchange_unique_id(uint8_t id[CLAPGPU_EXCHANGE_ID_BYTES])
{
    if (!id) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid u;
    if (g_rccl.get_uid(&u)) return CLAPGPU_ERR_UNKNOWN;
    memcpy(id, &u, sizeof(u));
    return CLAPGPU_OK;
}

extern "C" int clapgpu_exchange_create(clapgpu_exchange **out, const uint8_t id[CLAPGPU_EXCHANGE_ID_BYTES], int rank, int world)
{
    static_assert(CLAPGPU_EXCHANGE_ID_BYTES == sizeof(nccl_uid), "ncclUniqueId is 128 bytes");
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int rc = load_rccl();
    if (rc) return rc;
    clapgpu_exchange *x = static_cast<clapgpu_exchange *>(calloc(1, sizeof(*x)));
    if (!x) return CLAPGPU_ERR_NOMEM;
    nccl_uid u;
    memcpy(&u, id, sizeof(u));
    if (g_rccl.init_rank(&x->comm, world, u, rank)) { free(x); return CLAPGPU_ERR_INIT_FAILED; }
    x->rank = rank; x->world = world;
    *out = x;
    return CLAPGPU_OK;
}

// What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank) and which device this rank drives: a
// launcher gathers these from every rank and can then PROVE that N ranks on N distinct GPUs took part -- not N ranks
// that all landed on device 0, not a world of one.
extern "C" int clapgpu_exchange_info(const clapgpu_exchange *x, int *comm_ranks, int *comm_rank, char pci_bus_id[32])
{
    if (!x || !comm_ranks || !comm_rank || !pci_bus_id) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!g_rccl.comm_count || !g_rccl.comm_user_rank) return CLAPGPU_ERR_NOT_SUPPORTED;
    if (g_rccl.comm_count(x->comm, comm_ranks) || g_rccl.comm_user_rank(x->comm, comm_rank)) return CLAPGPU_ERR_UNKNOWN;
    int dev = 0;
    CLAPGPU_HIP(hipGetDevice(&dev));
    memset(pci_bus_id, 0, 32);
    CLAPGPU_HIP(hipDeviceGetPCIBusId(pci_bus_id, 32, dev));
    return CLAPGPU_OK;
}

extern "C" void clapgpu_exchange_destroy(clapgpu_exchange *x)
{
    if (!x) return;
    if (x->comm && g_rccl.destroy) g_rccl.destroy(x->comm);
    free(x);
}

extern "C" int clapgpu_exchange_visible(void *stream, clapgpu_exchange *x, const uint64_t *vis_mask, uint32_t n_pad,
                                        uint64_t *gathered_mask, uint32_t *visible, uint32_t *visible_count, void *scratch)
{
    if (!x || !vis_mask || !gathered_mask || (n_pad & 63u)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const size_t words = n_pad / 64;
    if (g_rccl.allgather(vis_mask, gathered_mask, words, NCCL_UINT64, x->comm, clapgpu::as_stream(stream)))
        return CLAPGPU_ERR_UNKNOWN;
    if (!visible) return CLAPGPU_OK;                             // the caller only wants the gathered mask
    if (!visible_count || !scratch) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if ((uint64_t)n_pad * (uint64_t)x->world > 0xffffffffull) return CLAPGPU_ERR_TOO_LARGE;
    // rank r's words sit at r * words: with equal padded shard sizes the gathered array IS the mask of the global range
    return clapgpu_visible_compact(stream, gathered_mask, nullptr, n_pad * (uint32_t)x->world, 0, visible, visible_count, scratch);
}

// Uneven shards (what clapgpu_shard_tile_range cuts from a real scene): every rank sends cap_pad / 64 words -- its own mask,
// zero beyond its n_pad -- and expands the gathered array with rank r's slot i numbered base[r] + i: scene-global ids, the
// identical ascending list on every rank.  base / n_pad: HOST arrays of `world` entries, the same on every rank
// (clapgpu_shard_bases).
extern "C" int clapgpu_exchange_visible_ranges(void *stream, clapgpu_exchange *x, const uint64_t *vis_mask, uint32_t cap_pad,
                                               const uint32_t *base, const uint32_t *n_pad, uint64_t *gathered_mask,
                                               uint32_t *visible, uint32_t *visible_count, void *scratch)
{
    if (!x || !vis_mask || !gathered_mask || !base || !cap_pad || (cap_pad & 63u)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (g_rccl.allgather(vis_mask, gathered_mask, cap_pad / 64, NCCL_UINT64, x->comm, clapgpu::as_stream(stream)))
        return CLAPGPU_ERR_UNKNOWN;
    if (!visible) return CLAPGPU_OK;                             // the caller only wants the gathered mask
    if (!visible_count || !scratch) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return clapgpu_visible_compact_ranges(stream, gathered_mask, (uint32_t)x->world, cap_pad, base, n_pad, visible, visible_count, scratch);
}

// Every rank's first global id and padded size for the cut clapgpu_shard_tile_range makes (rank r owns rows
// [tile_row_start[first_r], tile_row_start[end_r])), and the common capacity = the largest shard.  Host code.
extern "C" int clapgpu_shard_bases(const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t world, uint32_t *base, uint32_t *n_pad,
                                   uint32_t *cap_pad)
{
    if (!tile_row_start || !base || !n_pad || !world) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    uint32_t cap = 64;
    for (uint32_t r = 0; r < world; r++) {
        uint32_t t0, t1;
        int rc = clapgpu_shard_tile_range(tile_row_start, n_tiles, r, world, &t0, &t1);
        if (rc) return rc;
        const uint64_t b = (uint64_t)(tile_row_start[t0] - tile_row_start[0]) * 64, n = (uint64_t)(tile_row_start[t1] - tile_row_start[t0]) * 64;
        if (b + n > 0xffffffffull) return CLAPGPU_ERR_TOO_LARGE;
        base[r] = (uint32_t)b; n_pad[r] = (uint32_t)n;
        if (n > cap) cap = (uint32_t)n;
    }
    if (cap_pad) *cap_pad = cap;
    return CLAPGPU_OK;
}

// Contiguous tile ranges of (nearly) equal row count: whole tiles = whole subtrees stay on one rank, so the update needs
// no collective.  tile_row_start: n_tiles + 1 ascending row offsets (host).
extern "C" int clapgpu_shard_tile_range(const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t rank, uint32_t world,
                                        uint32_t *first_tile, uint32_t *end_tile)
{
    if (!tile_row_start || !first_tile || !end_tile || world == 0 || rank >= world) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint64_t total = tile_row_start[n_tiles] - tile_row_start[0];
    uint32_t cut[2];
    for (int k = 0; k < 2; k++) {
        const uint32_t r = rank + k;
        if (r == 0) { cut[k] = 0; continue; }
        if (r == world) { cut[k] = n_tiles; continue; }
        // the first tile whose start reaches total * r / world (compared without the division: exact)
        uint32_t lo = 0, hi = n_tiles;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if ((uint64_t)(tile_row_start[mid] - tile_row_start[0]) * world < total * r) lo = mid + 1;
            else hi = mid;
        }
        cut[k] = lo;
    }
    *first_tile = cut[0];
    *end_tile = cut[1] > cut[0] ? cut[1] : cut[0];
    return CLAPGPU_OK;
}
