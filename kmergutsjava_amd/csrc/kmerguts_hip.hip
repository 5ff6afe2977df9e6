// kmerguts_hip.hip -- host side of libkmerguts_hip.so (C ABI in include/kmerguts_hip.h).
//
// Replaces, for a batch of sequences, the reference's run() body between readFasta and the
// report printers (KGJ:776-816): prepareQuery/addKmers, the query sort, lookup and
// gatherHits/processSetOfHits.  Everything runs on one HIP stream owned by the table object;
// scratch and results come from a per-table cache of device blocks (DevCache) so that repeated
// scans reuse the same HBM.
#include "kg_device.hpp"
#include "kg_aggregate.hpp"
#include "kg_partition.hpp"
#include "kg_order.hpp"

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(KG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

// Device-memory cache of one table object.  Every scan ends with a stream synchronisation, and
// blocks are handed back only when the stream is idle, so a freed block can be reused by the next
// request without any ordering concern.  Keeps the working set of repeated scans resident in HBM
// (no hipMalloc/hipFree in the steady state).
struct DevCache {
    std::mutex mu;
    std::multimap<size_t, void *> free_;
    std::unordered_map<void *, size_t> live;

    static size_t round_up(size_t b)
    {
        if (b < 256) return 256;
        size_t g = b >= (8u << 20) ? (2u << 20) : 256;       // 2 MiB granules for large blocks
        return (b + g - 1) / g * g;
    }
    hipError_t get(void **p, size_t bytes)
    {
        bytes = round_up(bytes);
        std::lock_guard<std::mutex> lk(mu);
        auto it = free_.lower_bound(bytes);
        if (it != free_.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
            *p = it->second;
            live[*p] = it->first;
            free_.erase(it);
            return hipSuccess;
        }
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {
            // give cached blocks back to the driver and retry once
            for (auto &kv : free_) (void)hipFree(kv.second);
            free_.clear();
            e = hipMalloc(p, bytes);
            if (e != hipSuccess) return e;
        }
        live[*p] = bytes;
        return hipSuccess;
    }
    void put(void *p)
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) return;
        free_.emplace(it->second, p);
        live.erase(it);
    }
    size_t live_bytes()
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t n = 0;
        for (auto &kv : live) n += kv.second;
        return n;
    }
    void release_all()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : free_) (void)hipFree(kv.second);
        for (auto &kv : live) (void)hipFree(kv.first);
        free_.clear();
        live.clear();
    }
};

// Pinned host blocks for the result views.  hipHostMalloc / hipHostFree cost ~0.1-0.2 ms for a small block (more than
// a small scan) and ~0.5 s for the 880 MB of hit records of a 1 Gbp batch (page pinning: the copy itself takes 20 ms at
// PCIe rate), so blocks are kept for the next result: up to kKeepTotal bytes, largest dropped first.
struct PinCache {
    std::mutex mu;
    std::multimap<size_t, void *> free_;
    std::unordered_map<void *, size_t> live;
    size_t kept = 0;
    static constexpr size_t kKeepTotal = 6ull << 30;

    hipError_t get(void **p, size_t bytes)
    {
        bytes = bytes < 4096 ? 4096 : (bytes + 4095) / 4096 * 4096;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_.lower_bound(bytes);
            if (it != free_.end() && it->first <= 2 * bytes + (1u << 16)) {
                *p = it->second;
                live[*p] = it->first;
                kept -= it->first;
                free_.erase(it);
                return hipSuccess;
            }
        }
        hipError_t e = hipHostMalloc(p, bytes);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        live[*p] = bytes;
        return hipSuccess;
    }
    void put(void *p)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(p);
            if (it == live.end()) return;
            const size_t bytes = it->second;
            live.erase(it);
            free_.emplace(bytes, p);
            kept += bytes;
            while (kept > kKeepTotal && !free_.empty()) {          // largest first
                auto big = std::prev(free_.end());
                kept -= big->first;
                drop.push_back(big->second);
                free_.erase(big);
            }
        }
        for (void *d : drop) (void)hipHostFree(d);
    }
    void release_all()
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &kv : free_) (void)hipHostFree(kv.second);
        for (auto &kv : live) (void)hipHostFree(kv.first);
        free_.clear();
        live.clear();
        kept = 0;
    }
};

}  // namespace

constexpr uint64_t kHbitsMaxSlots = 1ull << 26;     // tables up to this many slots get the bit-per-slot digest (8 MB of bits)

struct kg_table {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // partitioned scan: tag pass of chunk c while chunk c+1 is scattered (stream)
    hipStream_t stream3 = nullptr;      // ... and while chunk c-1 is verified and placed
    hipStream_t ostream[4] = {};        // ordering streams (KG_ORDER_STREAMS), lowest priority: queues of their own
    hipEvent_t pev[48] = {};            // [2c] chunk c scattered, [2c+1] chunk c tag-probed (c < 8); [16],[17],[18] fork / joins;
                                        // [20+c] chunk c verified
    bool own_entries = false;
    uint8_t *d_entries = nullptr;
    uint8_t *d_tags = nullptr;
    uint8_t *d_bidx = nullptr;          // byte home index (kg_device.hpp, build_bidx_kernel): 1 byte per slot, limit + 64 bytes
    bool bidx_exact = false;            // every quotient < 19: its classes are quotients
    uint32_t *d_hbits = nullptr;        // one bit per slot: the byte above is not 0 (tables of at most kHbitsMaxSlots slots: the direct kernel's prefilter)
    uint64_t tail_start = 0;            // first slot of the occupied run that ends at the end of the record stream
    int64_t num_sigs = 0, entry_size = 0, version = 0;
    uint64_t limit = 0;          // complete 24-byte records present
    uint64_t magic = 0;          // floor(2^64 / num_sigs)
    uint32_t m35 = 0;            // floor(2^35 / num_sigs) when 64 <= num_sigs < 2^31 (kg::split_fast), else 0
    uint64_t occupied = 0;
    double stage_ratio = 1.0 / 16;   // staging records per window, grown to the high-water mark
    size_t scatter_lds[2] = {0, 0};  // dynamic LDS the scatter kernel (DNA / protein) has been allowed so far
    size_t hist_lds = 48 * 1024;     // ... and the hit histogram kernel (kg_order.hpp)
    size_t place_lds[2] = {48 * 1024, 48 * 1024};   // ... and group_place_kernel<DNA / AA>
    hipEvent_t ev[8] = {};
    // Pinned host words for the few counters a scan reads back (a hipMemcpyAsync to pageable memory blocks the host per
    // copy; to pinned memory it does not): [0..47] d_pc, [48..79] d_ovfc (as 64 x u32), [80..87] d_totals, [88] CALL total
    uint64_t *h_pin = nullptr;
    std::atomic<int> busy{0};    // a kg_scan* is in flight on this table (its streams, events and pinned words are per table)
    uint32_t fail_alloc_at = 0, alloc_count = 0;   // test hook KG_TEST_FAIL_ALLOC (include/kmerguts_hip.h)
    DevCache cache;
    PinCache pins;
};

struct kg_result {
    kg_table *tab = nullptr;
    bool own_tab = false;        // kg_aggregate_hits: the result owns a table-less context (stream + block caches)
    kg_stats st = {};
    uint32_t per = 6;
    // device
    kg_hit *d_hits = nullptr;
    int64_t *d_chs = nullptr;
    kg_call *d_calls = nullptr;
    int64_t *d_ccs = nullptr;
    kg_otu *d_otu = nullptr;
    uint8_t *d_ev = nullptr, *d_tail_ev = nullptr;   // KG_EV_* per hit / per container
    uint32_t *d_hit_slots = nullptr;                 // KG_F_PROGRESS: the slot every hit was found at
    bool has_progress = false;
    kg_progress progress = {};
    // host copies (lazy), in pinned memory so the copy runs at PCIe rate
    void *h_hits = nullptr, *h_chs = nullptr, *h_ccs = nullptr, *h_calls = nullptr, *h_otu = nullptr, *h_ev = nullptr,
         *h_tail_ev = nullptr, *h_hit_slots = nullptr;
};

namespace {

uint32_t env_u32(const char *name, uint32_t dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    long x = strtol(v, &end, 10);
    return (end != v && x >= 0) ? (uint32_t)x : dflt;
}

// The two test hooks (KG_TEST_TINY_LISTS, KG_TEST_FAIL_ALLOC; include/kmerguts_hip.h) are read only when the process opted in
// with KG_ENABLE_TEST_HOOKS=1 -- looked at ONCE, at the first scan: a stray KG_TEST_* variable in a server's environment
// does nothing.
uint32_t test_hook(const char *name)
{
    static const bool enabled = env_u32("KG_ENABLE_TEST_HOOKS", 0u) != 0;
    return enabled ? env_u32(name, 0u) : 0u;
}

int dalloc(kg_table *t, void **p, size_t bytes)
{
    if (t->fail_alloc_at && ++t->alloc_count == t->fail_alloc_at)
        return fail(KG_ERR_NOMEM, "device allocation failed: KG_TEST_FAIL_ALLOC test hook");
    hipError_t e = t->cache.get(p, bytes);
    if (e != hipSuccess) return fail(KG_ERR_NOMEM, std::string("device allocation failed: ") + hipGetErrorString(e));
    return KG_OK;
}

// Only call while the table's stream is idle (see DevCache).
void dfree(kg_table *t, void *p)
{
    if (p) t->cache.put(p);
}

int table_finish(kg_table *t)
{
    // tag array + occupancy count: one streaming pass over the records
    HIP_TRY(hipSetDevice(t->device));
    // the records may have been produced on another stream (kg_table_from_device): the library's
    // stream is non-blocking, so wait for everything the device has been given so far
    HIP_TRY(hipDeviceSynchronize());
    unsigned __int128 one = 1;
    if (t->num_sigs == 1) t->magic = ~0ull;
    else t->magic = (uint64_t)((one << 64) / (unsigned __int128)(uint64_t)t->num_sigs);
    t->m35 = (t->num_sigs >= 64 && t->num_sigs < (1ll << 31)) ? (uint32_t)((1ull << 35) / (uint64_t)t->num_sigs) : 0u;
    uint64_t n_tags = t->limit + kg::kTagPad;
    HIP_TRY(hipMalloc((void **)&t->d_tags, n_tags));
    unsigned long long *d_occ = nullptr;
    HIP_TRY(hipMalloc((void **)&d_occ, 16));
    HIP_TRY(hipMemsetAsync(d_occ, 0, 16, t->stream));
    uint64_t want = (n_tags + 255) / 256;
    uint32_t grid = (uint32_t)(want < 256ull * 16 ? (want ? want : 1) : 256ull * 16);
    hipLaunchKernelGGL(kg::build_tags_kernel, dim3(grid), dim3(256), 0, t->stream, t->d_entries, t->limit, n_tags,
                       (uint64_t)t->num_sigs, t->magic, t->d_tags, d_occ);
    HIP_TRY(hipGetLastError());
    // the byte home index: what the tag pass probes instead of the tags, for every table the scatter pass applies to
    // (KG_BIDX=0 switches it off per scan, not here: a table outlives the environment it was opened in)
    t->bidx_exact = (uint64_t)KG_MAX_ENCODED / (uint64_t)t->num_sigs + 1 <= kg::kBidxClasses;
    if (t->m35 != 0 && t->limit > 0) {
        const uint64_t n_bidx = t->limit + kg::kTagPad;
        HIP_TRY(hipMalloc((void **)&t->d_bidx, n_bidx));
        const uint64_t wantb = (n_bidx + 255) / 256;
        hipLaunchKernelGGL(kg::build_bidx_kernel, dim3((uint32_t)std::min<uint64_t>(wantb, 256ull * 32)), dim3(256), 0, t->stream,
                           t->d_entries, t->d_tags, t->limit, n_bidx, (uint64_t)t->num_sigs, t->magic, t->d_bidx);
        HIP_TRY(hipGetLastError());
        // ... and, for tables whose bits stay in an XCD's L2 or close to it, its one-bit-per-slot digest: the direct kernel asks it
        // first (scan_kernel).  2^26 slots = 8 MB of bits: the gather rate there is still twice that of a tag array eight times
        // the size (profiles/r01_gather_ceiling_small_tables.jsonl).
        if (n_bidx <= kHbitsMaxSlots) {
            const uint64_t n_words = (n_bidx + 31) / 32;
            HIP_TRY(hipMalloc((void **)&t->d_hbits, n_words * 4));
            hipLaunchKernelGGL(kg::build_hbits_kernel, dim3((uint32_t)std::min<uint64_t>((n_words + 255) / 256, 256ull * 32)), dim3(256), 0, t->stream,
                               t->d_bidx, n_bidx, t->d_hbits, n_words);
            HIP_TRY(hipGetLastError());
        }
    }
    unsigned long long occ[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(occ, d_occ, 16, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    HIP_TRY(hipFree(d_occ));
    t->occupied = occ[0];
    t->tail_start = occ[1];
    for (auto &e : t->pev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&t->stream3, hipStreamNonBlocking));
    return KG_OK;
}

int table_new(int device, kg_table **out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(KG_ERR_DEVICE, "no HIP device: libkmerguts_hip needs an MI355X (gfx950) GPU; there is no CPU path");
    if (device < 0 || device >= ndev) return fail(KG_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    kg_table *t = new (std::nothrow) kg_table();
    if (!t) return fail(KG_ERR_NOMEM, "out of host memory");
    t->device = device;
    hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&t->h_pin, 96 * 8);
    for (auto &ev : t->ev)
        if (e == hipSuccess) e = hipEventCreate(&ev);               // (a table-less context of kg_aggregate_hits uses them too)
    if (e != hipSuccess) { kg_table_close(t); return fail(KG_ERR_DEVICE, std::string("hipStreamCreate / hipEventCreate: ") + hipGetErrorString(e)); }
    *out = t;
    return KG_OK;
}

int64_t rd_i64le(const uint8_t *b)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
    return (int64_t)v;
}

int parse_header(const uint8_t *hdr, kg_table *t)
{
    // readKmerTableHeader, KGJ:933-935
    t->num_sigs = rd_i64le(hdr);
    t->entry_size = rd_i64le(hdr + 8);
    t->version = rd_i64le(hdr + 16);      // never checked by the reference (KGJ:97 VERSION unused)
    if (t->num_sigs <= 0) return fail(KG_ERR_FORMAT, "kmer table header: numSigs <= 0");
    if (t->entry_size != KG_TABLE_ENTRY_SIZE)
        return fail(KG_ERR_FORMAT, "kmer table header: entrySize != 24 (the reference reads 24-byte records, KGJ:995-999)");
    return KG_OK;
}

}  // namespace

extern "C" {

const char *kg_last_error(void) { return g_err.c_str(); }
const char *kg_version(void) { return "libkmerguts_hip 0.1.0 gfx950"; }

int kg_table_from_memory(const void *image, size_t nbytes, int device, kg_table **out)
{
    if (!image || !out) return fail(KG_ERR_ARG, "null argument");
    if (nbytes < 24) return fail(KG_ERR_FORMAT, "kmer table image shorter than its 24-byte header");
    kg_table *t = nullptr;
    int rc = table_new(device, &t);
    if (rc) return rc;
    rc = parse_header((const uint8_t *)image, t);
    if (rc) { kg_table_close(t); return rc; }
    t->limit = (nbytes - 24) / KG_TABLE_ENTRY_SIZE;       // a trailing partial record is an EOF for the reference
    size_t bytes = (size_t)t->limit * KG_TABLE_ENTRY_SIZE;
    t->own_entries = true;
    hipError_t e = hipMalloc((void **)&t->d_entries, bytes ? bytes : 256);
    if (e != hipSuccess) { kg_table_close(t); return fail(KG_ERR_NOMEM, std::string("hipMalloc(table): ") + hipGetErrorString(e)); }
    if (bytes) {
        e = hipMemcpy(t->d_entries, (const uint8_t *)image + 24, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { kg_table_close(t); return fail(KG_ERR_DEVICE, std::string("hipMemcpy(table): ") + hipGetErrorString(e)); }
    }
    rc = table_finish(t);
    if (rc) { kg_table_close(t); return rc; }
    *out = t;
    return KG_OK;
}

// gzip members are inflated by one zlib stream (a gzip stream has no block index: it cannot be cut for several
// threads); what can overlap does: the inflate of piece k+1 with the upload of piece k, and no host copy of the
// table is ever held (the reference's GZIPInputStream is a stream too, KGJ:749-753, 927-929).
static int open_gz(const char *path, int device, kg_table **out)
{
    gzFile g = gzopen(path, "rb");
    if (!g) return fail(KG_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
    gzbuffer(g, 1u << 20);
    uint8_t hdr[24];
    if (gzread(g, hdr, 24) != 24) { gzclose(g); return fail(KG_ERR_FORMAT, "kmer table file shorter than its 24-byte header"); }
    kg_table *t = nullptr;
    int rc = table_new(device, &t);
    if (rc) { gzclose(g); return rc; }
    rc = parse_header(hdr, t);
    if (rc) { gzclose(g); kg_table_close(t); return rc; }
    // the header says how many records to expect; a stream that holds more keeps being read by the reference, so the
    // device buffer grows when it has to
    size_t cap = (size_t)t->num_sigs * KG_TABLE_ENTRY_SIZE;
    if (cap < 256) cap = 256;
    t->own_entries = true;
    hipError_t e = hipMalloc((void **)&t->d_entries, cap);
    if (e != hipSuccess) { gzclose(g); kg_table_close(t); return fail(KG_ERR_NOMEM, std::string("hipMalloc(table): ") + hipGetErrorString(e)); }
    const size_t CH = 64u << 20;
    uint8_t *pin[2] = {nullptr, nullptr};
    hipEvent_t done[2];
    bool ok = hipHostMalloc((void **)&pin[0], CH) == hipSuccess && hipHostMalloc((void **)&pin[1], CH) == hipSuccess &&
              hipEventCreate(&done[0]) == hipSuccess && hipEventCreate(&done[1]) == hipSuccess;
    size_t at = 0;
    int which = 0;
    bool used[2] = {false, false};
    std::string why;
    while (ok) {
        if (used[which]) ok = hipEventSynchronize(done[which]) == hipSuccess;
        if (!ok) break;
        size_t n = 0;
        while (n < CH) {                                   // gzread takes an unsigned int
            const int got = gzread(g, pin[which] + n, (unsigned)std::min<size_t>(CH - n, 1u << 30));
            if (got < 0) { int en = 0; why = gzerror(g, &en); ok = false; break; }
            if (got == 0) break;
            n += (size_t)got;
        }
        if (!ok || n == 0) break;
        if (at + n > cap) {
            size_t ncap = std::max(at + n, cap + cap / 2);
            uint8_t *bigger = nullptr;
            ok = hipStreamSynchronize(t->stream) == hipSuccess && hipMalloc((void **)&bigger, ncap) == hipSuccess &&
                 hipMemcpy(bigger, t->d_entries, at, hipMemcpyDeviceToDevice) == hipSuccess;
            if (!ok) { if (bigger) (void)hipFree(bigger); why = "out of device memory for a table longer than its header says"; break; }
            (void)hipFree(t->d_entries);
            t->d_entries = bigger; cap = ncap;
        }
        ok = hipMemcpyAsync(t->d_entries + at, pin[which], n, hipMemcpyHostToDevice, t->stream) == hipSuccess &&
             hipEventRecord(done[which], t->stream) == hipSuccess;
        used[which] = true;
        at += n;
        which ^= 1;
    }
    if (ok) ok = hipStreamSynchronize(t->stream) == hipSuccess;
    gzclose(g);
    if (pin[0]) (void)hipHostFree(pin[0]);
    if (pin[1]) (void)hipHostFree(pin[1]);
    (void)hipEventDestroy(done[0]);
    (void)hipEventDestroy(done[1]);
    if (!ok) { kg_table_close(t); return fail(KG_ERR_IO, "inflating/uploading the kmer table failed" + (why.empty() ? std::string() : ": " + why)); }
    t->limit = at / KG_TABLE_ENTRY_SIZE;                  // a trailing partial record is an EOF for the reference
    rc = table_finish(t);
    if (rc) { kg_table_close(t); return rc; }
    *out = t;
    return KG_OK;
}

int kg_table_open(const char *path, int device, kg_table **out)
{
    if (!path || !out) return fail(KG_ERR_ARG, "null argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(KG_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
    uint8_t hdr[24];
    const size_t got_hdr = fread(hdr, 1, 24, f);
    if (got_hdr >= 2 && hdr[0] == 0x1f && hdr[1] == 0x8b) {          // gzip magic: kmer.table.mem_map.gz
        fclose(f);
        return open_gz(path, device, out);
    }
    if (got_hdr != 24) { fclose(f); return fail(KG_ERR_FORMAT, "kmer table file shorter than its 24-byte header"); }
    if (fseeko(f, 0, SEEK_END) != 0) { fclose(f); return fail(KG_ERR_IO, "fseek failed"); }
    off_t fsz = ftello(f);
    fclose(f);
    kg_table *t = nullptr;
    int rc = table_new(device, &t);
    if (rc) return rc;
    rc = parse_header(hdr, t);
    if (rc) { kg_table_close(t); return rc; }
    t->limit = (uint64_t)(fsz - 24) / KG_TABLE_ENTRY_SIZE;
    size_t bytes = (size_t)t->limit * KG_TABLE_ENTRY_SIZE;
    t->own_entries = true;
    hipError_t e = hipMalloc((void **)&t->d_entries, bytes ? bytes : 256);
    if (e != hipSuccess) { kg_table_close(t); return fail(KG_ERR_NOMEM, std::string("hipMalloc(table): ") + hipGetErrorString(e)); }
    // Several reader threads pread() disjoint 32 MiB pieces of the file into their own pinned buffers (two each) and
    // hand them to the copy engine: one thread's read() runs at the page cache's single-core memcpy rate (~5 GB/s),
    // a 33.6 GB table should load at what the PCIe link takes.
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { kg_table_close(t); return fail(KG_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno)); }
    const size_t CH = 32u << 20;
    const size_t n_pieces = (bytes + CH - 1) / CH;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_thr = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(hw ? hw : 4), n_pieces}));
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    std::mutex err_mu;
    std::string why;
    auto worker = [&]() {
        if (hipSetDevice(device) != hipSuccess) { ok = false; return; }
        hipStream_t s = nullptr;
        uint8_t *pin[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        bool good = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
                    hipHostMalloc((void **)&pin[0], CH) == hipSuccess && hipHostMalloc((void **)&pin[1], CH) == hipSuccess &&
                    hipEventCreate(&done[0]) == hipSuccess && hipEventCreate(&done[1]) == hipSuccess;
        bool used[2] = {false, false};
        int which = 0;
        while (good && ok.load()) {
            const size_t k = next.fetch_add(1);
            if (k >= n_pieces) break;
            const size_t at = k * CH, n = std::min(CH, bytes - at);
            if (used[which]) good = hipEventSynchronize(done[which]) == hipSuccess;
            size_t got = 0;
            while (good && got < n) {
                const ssize_t r = pread(fd, pin[which] + got, n - got, (off_t)(24 + at + got));
                if (r <= 0) { std::lock_guard<std::mutex> lk(err_mu); why = "short read on kmer table file"; good = false; break; }
                got += (size_t)r;
            }
            if (!good) break;
            good = hipMemcpyAsync(t->d_entries + at, pin[which], n, hipMemcpyHostToDevice, s) == hipSuccess &&
                   hipEventRecord(done[which], s) == hipSuccess;
            used[which] = true;
            which ^= 1;
        }
        if (s) (void)hipStreamSynchronize(s);
        if (!good) ok = false;
        for (int i = 0; i < 2; i++) { if (pin[i]) (void)hipHostFree(pin[i]); if (done[i]) (void)hipEventDestroy(done[i]); }
        if (s) (void)hipStreamDestroy(s);
    };
    {
        std::vector<std::thread> pool;
        for (size_t i = 1; i < n_thr; i++) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    close(fd);
    if (!ok.load()) { kg_table_close(t); return fail(KG_ERR_IO, "reading/uploading the kmer table failed" + (why.empty() ? std::string() : ": " + why)); }
    rc = table_finish(t);
    if (rc) { kg_table_close(t); return rc; }
    *out = t;
    return KG_OK;
}

int kg_table_from_device(const void *d_entries, int64_t num_sigs, int device, kg_table **out)
{
    if (!d_entries || !out) return fail(KG_ERR_ARG, "null argument");
    if (num_sigs <= 0) return fail(KG_ERR_ARG, "num_sigs <= 0");
    kg_table *t = nullptr;
    int rc = table_new(device, &t);
    if (rc) return rc;
    t->num_sigs = num_sigs;
    t->entry_size = KG_TABLE_ENTRY_SIZE;
    t->version = 1;
    t->limit = (uint64_t)num_sigs;
    t->own_entries = false;
    t->d_entries = (uint8_t *)d_entries;
    rc = table_finish(t);
    if (rc) { kg_table_close(t); return rc; }
    *out = t;
    return KG_OK;
}

int kg_table_info(const kg_table *t, int64_t *num_sigs, int64_t *entry_size, int64_t *version, int64_t *occupied)
{
    if (!t) return fail(KG_ERR_ARG, "null table");
    if (num_sigs) *num_sigs = t->num_sigs;
    if (entry_size) *entry_size = t->entry_size;
    if (version) *version = t->version;
    if (occupied) *occupied = (int64_t)t->occupied;
    return KG_OK;
}

void kg_table_close(kg_table *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->own_entries && t->d_entries) (void)hipFree(t->d_entries);
    if (t->d_tags) (void)hipFree(t->d_tags);
    if (t->d_bidx) (void)hipFree(t->d_bidx);
    if (t->d_hbits) (void)hipFree(t->d_hbits);
    t->cache.release_all();
    t->pins.release_all();
    if (t->h_pin) (void)hipHostFree(t->h_pin);
    for (auto &e : t->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : t->pev)
        if (e) (void)hipEventDestroy(e);
    if (t->stream2) { (void)hipStreamSynchronize(t->stream2); (void)hipStreamDestroy(t->stream2); }
    if (t->stream3) { (void)hipStreamSynchronize(t->stream3); (void)hipStreamDestroy(t->stream3); }
    for (auto &os : t->ostream) if (os) { (void)hipStreamSynchronize(os); (void)hipStreamDestroy(os); }
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
}

void kg_result_free(kg_result *r)
{
    if (!r) return;
    kg_table *t = r->tab;
    if (t) {
        // a result is only handed out after its scan has synchronised the stream
        dfree(t, r->d_hits); dfree(t, r->d_chs); dfree(t, r->d_calls); dfree(t, r->d_ccs); dfree(t, r->d_otu);
        dfree(t, r->d_ev); dfree(t, r->d_tail_ev); dfree(t, r->d_hit_slots);
    }
    for (void *h : {r->h_hits, r->h_chs, r->h_ccs, r->h_calls, r->h_otu, r->h_ev, r->h_tail_ev, r->h_hit_slots})
        if (h) { if (t) t->pins.put(h); else (void)hipHostFree(h); }
    if (t && r->own_tab) kg_table_close(t);
    delete r;
}

}  // extern "C"

namespace {

// exclusive prefix sum of d_in[n] -> d_out[n], total -> d_total (device uint64)
int prefix_sum(kg_table *t, const uint32_t *d_in, uint64_t n, uint32_t *d_out, uint64_t *d_partial, uint64_t *d_total,
               hipStream_t stream = nullptr)
{
    if (!stream) stream = t->stream;
    uint32_t nb = (uint32_t)((n + kg::kScanChunk - 1) / kg::kScanChunk);
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(kg::scan_partials_kernel, dim3(nb), dim3(kg::kScanThreads), 0, stream, d_in, n, d_partial);
    hipLaunchKernelGGL(kg::scan_top_kernel, dim3(1), dim3(kg::kScanThreads), 0, stream, d_partial, nb, d_total);
    hipLaunchKernelGGL(kg::scan_final_kernel, dim3(nb), dim3(kg::kScanThreads), 0, stream, d_in, n, d_partial, d_out);
    HIP_TRY(hipGetLastError());
    return KG_OK;
}

struct Scratch {
    kg_table *t;
    std::vector<void *> ptrs;
    explicit Scratch(kg_table *tt) : t(tt) {}
    ~Scratch()
    {
        (void)hipStreamSynchronize(t->stream);      // blocks go back to the cache only when both streams are idle
        if (t->stream2) (void)hipStreamSynchronize(t->stream2);
        if (t->stream3) (void)hipStreamSynchronize(t->stream3);
        for (auto &os : t->ostream) if (os) (void)hipStreamSynchronize(os);
        for (void *p : ptrs) dfree(t, p);
    }
    void adopt(void *p) { ptrs.push_back(p); }
    template <typename T> int get(T **p, size_t count)
    {
        void *v = nullptr;
        int rc = dalloc(t, &v, count * sizeof(T));
        if (rc) return rc;
        ptrs.push_back(v);
        *p = (T *)v;
        return KG_OK;
    }
};

// gatherHits / processSetOfHits / the OTU buffer (KGJ:385-524) over res->d_hits + res->d_chs: fills the CALL, OTU and event
// arrays of res.  d_partial: prefix-sum scratch for n_cont items, d_totals[4]: the CALL total.  otu_init (device, one record
// per sequence, or null): the oICounts buffers the sequences start with (kg_aggregate_hits; the scan starts them empty).
// Everything is enqueued on t->stream and nothing is waited for: calls[] is allocated for the most CALLs n_hits records can
// make (n_hits / minHits), so the host does not need the CALL total before the records are compacted; the total arrives in
// t->h_pin[kPinCalls] once the caller has synchronised the stream.
constexpr int kPinCalls = 88, kPinPieces = 89;
int aggregate_stage(kg_table *t, const kg_params *p, kg_result *res, Scratch &sc, int64_t n_seqs, uint64_t n_cont, uint64_t n_hits,
                    uint32_t PER, uint64_t *d_partial, uint64_t *d_totals, const kg_otu *d_otu_init, bool allow_pieces)
{
    int rc;
    {
        kg::AggParams ap;
        ap.min_hits = p->min_hits; ap.min_weighted_hits = p->min_weighted_hits;
        ap.max_gap = p->max_gap; ap.order_constraint = p->order_constraint ? 1 : 0;
        uint32_t *d_ccnt = nullptr, *d_coff = nullptr, *d_first = nullptr;
        kg_call *d_staged = nullptr;
        uint8_t *d_vote = nullptr;
        if ((rc = dalloc(t, (void **)&res->d_ev, n_hits))) return rc;
        if ((rc = dalloc(t, (void **)&res->d_tail_ev, n_cont))) return rc;
        uint8_t *d_acc = res->d_ev;
        if ((rc = sc.get(&d_ccnt, n_cont))) return rc;
        if ((rc = sc.get(&d_first, n_cont))) return rc;
        if ((rc = sc.get(&d_coff, n_cont))) return rc;
        if ((rc = sc.get(&d_vote, n_hits))) return rc;
        // a hit votes for at most one CALL and a CALL needs >= minHits voters: the CALLs of a unit (a container, or a piece of a
        // long one) that starts at record b and ends before record e fit in [b / minHits, e / minHits) of the staging array
        if ((rc = sc.get(&d_staged, (size_t)(n_hits / (uint64_t)p->min_hits + 1)))) return rc;
        if ((rc = dalloc(t, (void **)&res->d_ccs, (n_cont + 1) * 8))) return rc;
        if ((rc = dalloc(t, (void **)&res->d_otu, (size_t)(n_seqs ? n_seqs : 1) * sizeof(kg_otu)))) return rc;
        // Long containers in pieces that start behind a gap > maxGap (kg_aggregate.hpp): exact when no -O (with it the gap
        // is measured from the last ACCEPTED record) and position + maxGap cannot wrap (the caller vouches for positions
        // < 2^30).  KG_AGG_BLOCK_SHIFT: log2 of the records per block (at most one piece start per block; tests lower it).
        const uint32_t pshift = std::min(20u, std::max(6u, env_u32("KG_AGG_BLOCK_SHIFT", 9u)));
        const bool pieces = allow_pieces && !p->order_constraint && p->max_gap >= 0 && p->max_gap < (1 << 30) && n_cont &&
                            n_hits > (2ull << pshift) && env_u32("KG_AGG_PIECES", 1u) != 0;
        const uint32_t n_pblocks = pieces ? (uint32_t)((n_hits + (1ull << pshift) - 1) >> pshift) : 0u;
        uint32_t *d_pstart = nullptr, *d_pcnt = nullptr;
        uint8_t *d_before = nullptr, *d_ppair = nullptr;
        t->h_pin[kPinCalls] = 0;
        t->h_pin[kPinPieces] = 0;
        {   // clears: the containers' CALL totals (units add to them), the pieces' counts and hand-over bytes
            kg::ClearList cl;
            cl.n = 0;
            for (int k = 0; k < 8; k++) { cl.p[k] = nullptr; cl.words[k] = 0; }
            if (n_cont) { cl.p[cl.n] = d_ccnt; cl.words[cl.n++] = n_cont; }
            if (pieces) {
                if ((rc = sc.get(&d_pstart, (size_t)n_pblocks + 1))) return rc;
                if ((rc = sc.get(&d_pcnt, (size_t)n_pblocks + 1))) return rc;
                if ((rc = sc.get(&d_before, ((size_t)n_pblocks + 4) & ~(size_t)3))) return rc;
                if ((rc = sc.get(&d_ppair, ((size_t)n_pblocks + 4) & ~(size_t)3))) return rc;
                cl.p[cl.n] = d_pcnt; cl.words[cl.n++] = (uint64_t)n_pblocks + 1;
                cl.p[cl.n] = reinterpret_cast<uint32_t *>(d_before); cl.words[cl.n++] = ((uint64_t)n_pblocks + 4) / 4;
            }
            if (cl.n) {
                uint64_t most = 0;
                for (int k = 0; k < cl.n; k++) most = std::max(most, cl.words[k]);
                hipLaunchKernelGGL(kg::clear_many_kernel, dim3((uint32_t)std::min<uint64_t>(1024, most / 1024 + 1)), dim3(256), 0, t->stream, cl);
            }
        }
        if (pieces)
            hipLaunchKernelGGL(kg::piece_starts_kernel, dim3((n_pblocks + 3) / 4), dim3(256), 0, t->stream, res->d_hits, res->d_chs,
                               (uint32_t)n_hits, pshift, ap.max_gap, d_pstart, d_ppair, n_pblocks, env_u32("KG_AGG_PAIRS", 1u));
        // one wave per unit: the containers' first pieces (several consecutive containers per wave when there are millions of
        // them: short reads), then one per block of hits[] that a later piece may start in
        const uint32_t cpw = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, n_cont / (1u << 17)));
        const uint32_t n_cwaves = (uint32_t)((n_cont + cpw - 1) / cpw);
        if (n_cont) {
            hipLaunchKernelGGL(kg::calls_wave_kernel, dim3((n_cwaves + n_pblocks + 3) / 4), dim3(256), 0, t->stream, res->d_hits, res->d_chs,
                               (uint32_t)n_cont, ap, d_acc, d_vote, res->d_tail_ev, d_ccnt, d_first, d_staged, cpw, n_cwaves, d_pstart,
                               pshift, n_pblocks, d_pcnt, d_before, d_ppair);
            if (pieces)
                hipLaunchKernelGGL(kg::merge_before_kernel, dim3((n_pblocks + 255) / 256), dim3(256), 0, t->stream, d_pstart, d_ppair, d_before,
                                   n_pblocks, res->d_ev, (unsigned long long *)(d_totals + 6));
            HIP_TRY(hipGetLastError());
        }
        if ((rc = prefix_sum(t, d_ccnt, n_cont, d_coff, d_partial, d_totals + 4))) return rc;
        if (n_cont) HIP_TRY(hipMemcpyAsync(t->h_pin + kPinCalls, d_totals + 4, 8, hipMemcpyDeviceToHost, t->stream));
        if (pieces) HIP_TRY(hipMemcpyAsync(t->h_pin + kPinPieces, d_totals + 6, 8, hipMemcpyDeviceToHost, t->stream));
        if (n_seqs) {
            // the voters of all CALLs as one dense list of otuIndex values in record order, then the replay per sequence
            const uint32_t n_vchunks = (uint32_t)((n_hits + 63) / 64);
            uint32_t *d_vcnt = nullptr, *d_voff = nullptr;
            int32_t *d_vlist = nullptr;
            uint64_t *d_vpartial = nullptr;
            if ((rc = sc.get(&d_vcnt, (size_t)n_vchunks + 1))) return rc;
            if ((rc = sc.get(&d_voff, (size_t)n_vchunks + 1))) return rc;
            if ((rc = sc.get(&d_vlist, (size_t)n_hits + 1))) return rc;
            if ((rc = sc.get(&d_vpartial, (size_t)((n_vchunks + 1) / kg::kScanChunk + 2)))) return rc;
            if (n_hits) {
                const uint32_t vgrid = (uint32_t)((n_hits + 255) / 256);
                // (n_vchunks + 1 items: the kernel zeroes the entry behind the last chunk; its prefix is the total, read for
                //  "behind the last record")
                hipLaunchKernelGGL(kg::voter_count_kernel, dim3(vgrid), dim3(256), 0, t->stream, d_vote, (uint32_t)n_hits, d_vcnt);
                if ((rc = prefix_sum(t, d_vcnt, (uint64_t)n_vchunks + 1, d_voff, d_vpartial, d_totals + 7))) return rc;
                hipLaunchKernelGGL(kg::voter_scatter_kernel, dim3(vgrid), dim3(256), 0, t->stream, res->d_hits, d_vote, (uint32_t)n_hits, d_voff,
                                   d_vlist);
            } else {
                HIP_TRY(hipMemsetAsync(d_voff, 0, 4, t->stream));
            }
            const uint32_t spw = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, (uint64_t)n_seqs / (1u << 17)));
            hipLaunchKernelGGL(kg::otu_wave_kernel, dim3((uint32_t)((((uint64_t)n_seqs + spw - 1) / spw + 3) / 4)), dim3(256), 0, t->stream,
                               d_vlist, d_voff, d_vote, res->d_chs, (uint32_t)n_hits, (uint32_t)n_seqs, PER, res->d_otu, spw, d_otu_init);
        }
        hipLaunchKernelGGL(kg::call_starts_kernel, dim3((uint32_t)((n_cont + 1 + 255) / 256)), dim3(256), 0, t->stream, d_coff,
                           n_cont, d_totals + 4, res->d_ccs);
        if ((rc = dalloc(t, (void **)&res->d_calls, (size_t)(n_hits / (uint64_t)p->min_hits + 1) * sizeof(kg_call)))) return rc;
        if (n_cont) {
            if (n_cont < (1u << 17))
                hipLaunchKernelGGL((kg::compact_calls_kernel<64>), dim3((uint32_t)((n_cont * 64 + 255) / 256)), dim3(256), 0, t->stream,
                                   d_staged, res->d_chs, d_first, d_coff, (uint32_t)n_cont, (uint32_t)p->min_hits, res->d_calls,
                                   d_pstart, d_pcnt, pshift);
            else
                hipLaunchKernelGGL((kg::compact_calls_kernel<1>), dim3((uint32_t)((n_cont + 255) / 256)), dim3(256), 0, t->stream,
                                   d_staged, res->d_chs, d_first, d_coff, (uint32_t)n_cont, (uint32_t)p->min_hits, res->d_calls,
                                   d_pstart, d_pcnt, pshift);
        }
        HIP_TRY(hipGetLastError());
    }
    return KG_OK;
}

template <bool AA>
int scan_impl(kg_table *t, const kg_params *p, const uint8_t *d_seq, const uint8_t *h_seq /* host copy still to upload, or null */,
              const int64_t *offsets, int64_t n_seqs, kg_result *res)
{
    // h_seq != null: d_seq is an empty device buffer of offsets[n_seqs] bytes; the characters are uploaded here -- chunk
    // by chunk in front of each chunk's scatter pass (partitioned strategy: the upload of chunk c+1 runs while chunk c is
    // scanned), or in one piece
    bool seq_uploaded = h_seq == nullptr;
    auto upload = [&](int64_t a, int64_t b) -> int {
        if (h_seq && b > a) HIP_TRY(hipMemcpyAsync(const_cast<uint8_t *>(d_seq) + a, h_seq + a, (size_t)(b - a), hipMemcpyHostToDevice, t->stream));
        return KG_OK;
    };
    constexpr uint32_t PER = AA ? 1 : 6;
    const bool progress = (p->flags & KG_F_PROGRESS) != 0;
    const bool counters = (p->flags & KG_F_COUNTERS) != 0 || progress;       // the walks are noted by the counting kernels
    kg_stats &st = res->st;
    res->per = PER;

    // ---- host: window blocks per sequence (KGJ:912 trip counts) ----
    std::vector<uint32_t> ibase((size_t)n_seqs + 1);
    uint64_t nblocks = 0, windows = 0, residues = 0;
    int64_t longest = 0;                            // (record positions are below the length of their sequence)
    for (int64_t k = 0; k < n_seqs; k++) {
        int64_t L = offsets[k + 1] - offsets[k];
        if (L < 0) return fail(KG_ERR_ARG, "offsets must be non-decreasing");
        longest = std::max(longest, L);
        if (L > 0xFFFFFFF0ll) return fail(KG_ERR_LIMIT, "a single sequence longer than 2^32-16 characters");
        ibase[(size_t)k] = (uint32_t)nblocks;
        uint64_t nb;
        if (AA) {
            uint64_t nwin = L >= 9 ? (uint64_t)L - 8 : 0;       // i < len - 8
            windows += nwin;
            residues += (uint64_t)L;
            nb = (nwin + kg::kAaWinPerBlock - 1) / kg::kAaWinPerBlock;
        } else {
            uint64_t npos = L >= 24 ? (uint64_t)L - 23 : 0;     // forward positions that start a 24-base window
            windows += 2 * npos;
            for (int f = 0; f < 3; f++)
                if (L - f >= 3) residues += 2 * (uint64_t)((L - f) / 3);
            nb = (npos + kg::kDnaPosPerBlock - 1) / kg::kDnaPosPerBlock;
        }
        nblocks += nb;
        if (nblocks > 0x7FFFFFFFull / PER) return fail(KG_ERR_LIMIT, "batch too large: more than 2^31-1 window rows; split the batch");
    }
    ibase[(size_t)n_seqs] = (uint32_t)nblocks;
    if (windows > 0xFFFFFF00ull) return fail(KG_ERR_LIMIT, "batch too large: more than 2^32-256 windows; split the batch");
    const uint64_t n_rows = nblocks * PER;
    const uint64_t n_cont = (uint64_t)n_seqs * PER;
    st.n_seqs = n_seqs; st.n_containers = (int64_t)n_cont; st.n_blocks = (int64_t)nblocks;
    st.residues = (int64_t)residues; st.windows = (int64_t)windows;
    st.table_bytes = t->num_sigs * (int64_t)KG_TABLE_ENTRY_SIZE;

    Scratch sc(t);
    int rc;
    int64_t *d_off = nullptr; uint32_t *d_ibase = nullptr;
    if ((rc = sc.get(&d_off, (size_t)n_seqs + 1))) return rc;
    if ((rc = sc.get(&d_ibase, (size_t)n_seqs + 1))) return rc;
    HIP_TRY(hipMemcpyAsync(d_off, offsets, ((size_t)n_seqs + 1) * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(d_ibase, ibase.data(), ((size_t)n_seqs + 1) * 4, hipMemcpyHostToDevice, t->stream));

    kg::BlockDesc *d_blocks = nullptr;
    uint32_t *d_counts = nullptr, *d_offs = nullptr, *d_bsb = nullptr;
    uint64_t *d_partial = nullptr, *d_totals = nullptr;   // totals[0] hits, [1] cursor, [2] ctr_valid, [3] ctr_slots, [4] calls, [5] ran off (sticky)
    if ((rc = sc.get(&d_blocks, nblocks))) return rc;
    if ((rc = sc.get(&d_counts, n_rows))) return rc;
    if ((rc = sc.get(&d_offs, n_rows))) return rc;
    if ((rc = sc.get(&d_bsb, nblocks * 6))) return rc;      // one staging base per (block, row group)
    uint64_t max_scan = n_rows > n_cont ? n_rows : n_cont;
    if ((rc = sc.get(&d_partial, (size_t)(max_scan / kg::kScanChunk + 2)))) return rc;
    if ((rc = sc.get(&d_totals, 8))) return rc;
    HIP_TRY(hipMemsetAsync(d_totals, 0, 64, t->stream));

    if ((rc = dalloc(t, (void **)&res->d_chs, (n_cont + 1) * 8))) return rc;

    // KG_F_PROGRESS: the walks' summary (kg_device.hpp, Progress).  lo[f] = the smallest slot of tenth >= f, found with the
    // reference's own double arithmetic (KGJ:1018) around ceil(f * numSigs / 10) - 1
    kg::Progress *d_prog = nullptr;
    if (progress) {
        if (t->limit > 0xFFFFFFFFull) return fail(KG_ERR_UNSUPPORTED, "KG_F_PROGRESS: table streams of 2^32 records or more");
        if ((rc = sc.get(&d_prog, 1))) return rc;
        kg::Progress h;
        for (auto &x : h.first) x = ~0ull;
        h.last_plus1 = 0; h.first_beyond = ~0ull; h.walk_ran_off = 0;
        for (auto &x : h.found_upto) x = 0;
        h.kmers_found = 0;
        const double n = (double)t->num_sigs;
        auto tenth = [&](uint64_t s) { return (int)(10.0 * ((double)(s + 1) / n)); };
        for (int f = 0; f <= 10; f++) {
            const unsigned __int128 num = (unsigned __int128)(uint64_t)t->num_sigs * (unsigned)f;
            uint64_t s = (uint64_t)((num + 9) / 10);
            s = s > 3 ? s - 3 : 0;
            while (tenth(s) < f) s++;
            h.lo[f] = s;
        }
        HIP_TRY(hipMemcpyAsync(d_prog, &h, sizeof h, hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));                             // (h is a stack object)
    }
    HIP_TRY(hipEventRecord(t->ev[0], t->stream));
    if (nblocks) {
        hipLaunchKernelGGL(kg::build_blocks_kernel, dim3((uint32_t)((nblocks + 255) / 256)), dim3(256), 0, t->stream,
                           d_off, d_ibase, (uint32_t)n_seqs, (uint32_t)nblocks, d_blocks);
        HIP_TRY(hipGetLastError());
    }

    // ---- strategy: direct probing (every probe a random 128-byte line from HBM unless the tag array is
    //      L2-sized) or partitioned probing (queries bucketed by slot range first; kg_partition.hpp) ----
    uint64_t n_hits = 0;
    st.scan_launches = 0;
    uint32_t part_shift = 0, part_buckets = 0;
    bool use_part = false;
    {
        // bucket = 2^shift slots (= bytes of tags); at most kMaxBuckets buckets; quotient must fit 32 - shift bits
        uint32_t shift = env_u32("KG_PART_SHIFT", 21u);
        const uint64_t qmax = (uint64_t)KG_MAX_ENCODED / (uint64_t)t->num_sigs + 1;
        while (shift > 4 && qmax >= (1ull << (32 - shift))) shift--;     // small tables: large quotients, small buckets
        while (((t->limit + (1ull << shift) - 1) >> shift) > (uint64_t)kg::kMaxBuckets) shift++;
        // the scatter workgroup keeps a 128-byte buffer per bucket in LDS: at most 160 KiB with its encode scratch
        while (AA ? kg::scatter_lds_bytes<true>((uint32_t)((t->limit + (1ull << shift) - 1) >> shift)) > 160u * 1024
                  : kg::scatter_lds_bytes<false>((uint32_t)((t->limit + (1ull << shift) - 1) >> shift)) > 160u * 1024)
            shift++;
        // the scatter pass splits k-mers with kg::split_fast: 64 <= numSigs < 2^31
        // (and the tag / verify passes keep slots in 32 bits: a table FILE may be longer than numSigs, KGJ:964-999)
        const bool fits = shift < 32 && qmax < (1ull << (32 - shift)) && nblocks <= (1ull << 23) && t->m35 != 0 &&
                          t->limit < (1ull << 32) - 64;
        // Measured against the 33.6 GB table (profiles/r01_partition_path.md), whole scan incl. ordering, direct vs
        // partitioned: 1 Gbp 35.0 / 21.4 ms, 600 Mbp 21.8 / 13.6, 400 Mbp 14.6 / 9.4, 200 Mbp 7.0 / 5.4, 100 Mbp 3.6 / 3.4 (one chunk).
        // Small inputs and L2/MALL-sized tables stay on the direct kernel.
        // KG_PARTITION: 0 direct, 1 partitioned whenever possible, 2 (default) auto.
        const uint32_t mode = env_u32("KG_PARTITION", 2u);
        const bool worth = t->limit >= (64ull << 20) && windows >= (1ull << 27);
        use_part = fits && nblocks > 0 && (mode == 1 || (mode == 2 && worth));
        part_shift = shift;
        part_buckets = (uint32_t)((t->limit + (1ull << shift) - 1) >> shift);
    }
    bool part_done = false;
    if (use_part) do {
        constexpr uint32_t WIN = AA ? 64u : 384u;                                    // windows per block
        constexpr uint32_t kMaxChunks = 8;
        const uint32_t per_iter = kg::kScatterWaves;
        // The batch is cut into chunks of whole sequences.  Chunk c goes through scatter (stream), tag pass (stream2),
        // then verification and ordered placement (stream3) while the chunks behind it are scattered and probed: the scatter pass is LDS/issue-
        // bound with one 16-wave workgroup per CU, the tag pass is L2-bound with few registers and no LDS, verification
        // and placement wait on random HBM lines, so they share the CUs.  A chunk's hits are a contiguous range of
        // hits[] (whole sequences), chained by a device-side running total.
        uint32_t want = env_u32("KG_PART_CHUNKS", 4u);
        if (want < 1) want = 1;
        if (want > kMaxChunks) want = kMaxChunks;
        // How many: a pass has costs that do not shrink with the chunk, so small batches take few.  Measured with the wave
        // priorities in place (r04 c59; ms per scan in 1 / 2 / 3 / 4 chunks): 100 Mbp 2.50 / 2.46 / 2.72 / -, 125 Mbp 2.94 / 2.87 /
        // 3.17 / -, 250 Mbp 5.15 / 4.83 / 5.17 / -, 500 Mbp - / - / 8.63 / 9.0, 1 Gbp - / - / 16.0 / 15.1 (five: 15.45):
        // round(sqrt(blocks / 325 000)) but at least two, one below 450 000 blocks (~85 Mbp).  KG_PART_MIN_CHUNK_BLOCKS (tests) replaces the
        // rule by "as many as KG_PART_CHUNKS allows with at least that many blocks each".
        if (getenv("KG_PART_MIN_CHUNK_BLOCKS")) {
            const uint64_t min_chunk = std::max(1u, env_u32("KG_PART_MIN_CHUNK_BLOCKS", 600000u));
            while (want > 1 && nblocks / want < min_chunk) want--;
        } else {
            const uint32_t by_size = nblocks < 450000 ? 1u : std::max(2u, (uint32_t)std::lround(std::sqrt((double)nblocks / 325000.0)));
            want = std::min(want, std::max(1u, by_size));
        }
        std::vector<uint64_t> clo;                                                    // chunk c = blocks [clo[c], clo[c+1])
        std::vector<int64_t> cseq;                                                    //         = sequences [cseq[c], cseq[c+1])
        clo.push_back(0); cseq.push_back(0);
        // KG_PART_TAPER="30,30,25,15": chunk sizes in percent instead of equal chunks (tuning aid)
        std::vector<double> cum;
        if (const char *tp = getenv("KG_PART_TAPER")) {
            double acc = 0;
            for (const char *q = tp; *q;) {
                char *endp = nullptr;
                const double v = strtod(q, &endp);
                if (endp == q) break;
                acc += v; cum.push_back(acc);
                q = *endp == ',' ? endp + 1 : endp;
            }
            if (cum.size() >= 2 && cum.size() <= kMaxChunks && acc > 0) { for (auto &x : cum) x /= acc; want = (uint32_t)cum.size(); }
            else cum.clear();
        }
        for (uint32_t c = 1; c < want; c++) {
            const uint64_t target = cum.empty() ? nblocks * c / want : (uint64_t)((double)nblocks * cum[c - 1]);
            const auto it = std::lower_bound(ibase.begin(), ibase.end(), (uint32_t)target);        // a sequence start
            const uint64_t cut = *it;
            if (cut > clo.back() && cut < nblocks) { clo.push_back(cut); cseq.push_back((int64_t)(it - ibase.begin())); }
        }
        clo.push_back(nblocks); cseq.push_back(n_seqs);
        const uint32_t n_chunks_p = (uint32_t)clo.size() - 1;
        uint64_t max_chunk = 0;
        for (uint32_t c = 0; c < n_chunks_p; c++) max_chunk = std::max(max_chunk, clo[c + 1] - clo[c]);
        const uint64_t chunk_blocks = (max_chunk + per_iter - 1) / per_iter * per_iter;
        const double max_frac = (double)max_chunk / (double)nblocks;
        uint32_t n_wg = env_u32("KG_PART_WGS", 256u);
        if ((uint64_t)n_wg * per_iter > chunk_blocks) n_wg = (uint32_t)((chunk_blocks + per_iter - 1) / per_iter);
        const uint64_t blocks_per_wg = ((chunk_blocks + (uint64_t)n_wg * per_iter - 1) / ((uint64_t)n_wg * per_iter)) * per_iter;
        // region capacity: the mean if every window were valid and hashed uniformly, plus 6 sigma, in 16-entry groups
        const double mean = (double)blocks_per_wg * WIN / (double)part_buckets * (env_u32("KG_PART_SLACK", 100u) / 100.0);
        const uint64_t cap64 = ((uint64_t)(mean + 6.0 * std::sqrt(mean) + 32.0) + 15) / 16 * 16;
        // the scatter pass's address arithmetic is in 24-bit multiplies (region number x capacity): geometries beyond that
        // (one bucket and millions of blocks per scatter workgroup; not reachable with the default knobs) take the direct path
        if (cap64 >= (1ull << 24) || (uint64_t)part_buckets * n_wg >= (1ull << 24)) break;
        const uint32_t cap = (uint32_t)cap64;
        const uint64_t n_regions_total = (uint64_t)part_buckets * n_wg;               // per chunk
        // overflow list of one chunk (groups): an eighth of the regions' capacity (low-complexity sequence: 3 % of the
        // bases in homopolymer runs overflow ~5 % of the entries; beyond the list the scan falls back to direct probing)
        const uint32_t ovf_cap = env_u32("KG_PART_OVF_GROUPS", (uint32_t)std::min<uint64_t>(1u << 23, std::max<uint64_t>(65536, n_regions_total * cap / 16 / 8)));
        uint64_t *d_ent = nullptr, *d_ovf_ent = nullptr;
        uint32_t *d_fill = nullptr, *d_ovf_bucket = nullptr, *d_next = nullptr, *d_ovfc = nullptr;
        kg::RowGeo *d_geo = nullptr;         // per row: container and position of its first window (kg_order.hpp)
        uint64_t *d_pc = nullptr;            // [0..7] hit-list cursors, [8..15] candidate cursors, [16..24] base, [32..39] chunk totals
        // ordered placement (kg_order.hpp): groups of 2^gshift rows, at most kMaxGroups per chunk (8192 while 4096-row groups allow it)
        uint32_t gshift = 10;
        while (gshift < 12 && ((max_chunk * PER) >> gshift) + 2 > 8192) gshift++;
        const uint32_t groups_stride = (uint32_t)(((max_chunk * PER) >> gshift) + 2);      // a chunk's rows start anywhere inside a group
        if (groups_stride > kg::kMaxGroups) return fail(KG_ERR_LIMIT, "a chunk of the batch holds more than 2^26 window rows");
        uint32_t *d_ghist = nullptr, *d_gbase = nullptr, *d_gcur1 = nullptr, *d_gcur2 = nullptr, *d_gtile = nullptr;
        kg_hit *d_sortA = nullptr, *d_sortB = nullptr;
        if ((rc = sc.get(&d_ent, (size_t)(n_regions_total * cap * n_chunks_p)))) return rc;
        if ((rc = sc.get(&d_fill, (size_t)n_regions_total * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_ovf_ent, (size_t)ovf_cap * kg::kGroup * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_ovf_bucket, (size_t)ovf_cap * n_chunks_p))) return rc;
        const size_t next_stride = std::max<size_t>((size_t)part_buckets + 8, 256);   // tag pass: one hand-out counter per XCD group, 128 B apart
        if ((rc = sc.get(&d_next, next_stride * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_ovfc, 8 * kMaxChunks))) return rc;       // per chunk: [0] overflow groups, [1] low-complexity blocks
        uint32_t *d_lowc = nullptr;                                    // block numbers set aside by the scatter pass
        if ((rc = sc.get(&d_lowc, (size_t)nblocks + 1))) return rc;
        if ((rc = sc.get(&d_geo, (size_t)n_rows))) return rc;
        if ((rc = sc.get(&d_ghist, (size_t)groups_stride * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_gbase, (size_t)(groups_stride + 1) * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_gcur1, (size_t)(kg::kHDigits + 1) * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_gcur2, (size_t)groups_stride * n_chunks_p))) return rc;
        if ((rc = sc.get(&d_gtile, (size_t)(kg::kHDigits + 1) * n_chunks_p))) return rc;
        if (kg::group_place_lds(gshift, gshift == 10) > t->place_lds[AA ? 1 : 0]) {
            const size_t want_lds = kg::group_place_lds(gshift, gshift == 10);
            HIP_TRY(hipFuncSetAttribute((const void *)kg::group_place_kernel<AA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want_lds));
            t->place_lds[AA ? 1 : 0] = want_lds;
        }
        if (groups_stride * 4u > t->hist_lds) {
            HIP_TRY(hipFuncSetAttribute((const void *)kg::hit_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(groups_stride * 4u)));
            t->hist_lds = groups_stride * 4u;
        }
        if ((rc = sc.get(&d_pc, 48))) return rc;
        unsigned long long *d_ctr = (unsigned long long *)(d_totals + 2);
        // the tag pass on the byte home index instead of the tags (bucket_index_kernel) unless the scan counts the slots it
        // inspects (the walk the index avoids) or KG_BIDX=0
        const bool use_bidx = t->d_bidx != nullptr && !counters && env_u32("KG_BIDX", 1u) != 0;
        const size_t lds = kg::scatter_lds_bytes<AA>(part_buckets);
        if (t->scatter_lds[AA ? 1 : 0] < lds) {         // once per table (and geometry): the call costs tens of microseconds
            HIP_TRY(hipFuncSetAttribute((const void *)kg::part_scatter_kernel<AA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            t->scatter_lds[AA ? 1 : 0] = lds;
        }
        // Tag workgroups per CU.  How many of them run beside a scatter workgroup of the next chunk is decided by the SIMDs'
        // VGPRs (kg_partition.hpp, "Register budgets": two per CU since round 3, one before), the rest wait for the scatter
        // workgroup to leave; the hand-out is by ticket, so the count only decides how fast freed registers are taken up.
        // Round 2 (one tag wave per SIMD beside the scatter pass): 4 per CU 20.4 ms, 8 per CU 20.8 (profiles/r02_pipeline.md);
        // round 3 (two): 4 per CU 19.78 ms, 8 per CU 19.56, bench.py 20.5 -> 20.25 ms per step (profiles/r03_experiments.md).
        const uint32_t probe_grid = env_u32("KG_PROBE_GRID", 256u * 8u) & ~7u;
        // the byte-index pass: four workgroups per CU -- at 32 VGPRs they are the four waves per SIMD that fit beside a scatter
        // workgroup (4 x 96 + 4 x 32 = 512); with eight queued the stage is 0.4 ms slower (16.37 against 15.93 ms, r04 c04)
        const uint32_t index_grid = env_u32("KG_INDEX_GRID", 256u * 4u) & ~7u;
        // ... and the regions it takes per hand-out: regions expected to hold fewer than ~640 / ~320 entries (about 0.7 of the
        // mean the capacity was computed from is valid DNA) are handed out two / four at a time (bucket_index_kernel)
        // wave priorities (s_setprio) of the two passes that share the CUs: kg_device.hpp, set_wave_prio
        const uint32_t scatter_prio = std::min(3u, env_u32("KG_SCATTER_PRIO", 1u)), index_prio = std::min(3u, env_u32("KG_INDEX_PRIO", 2u)),
                       verify_prio = std::min(3u, env_u32("KG_VERIFY_PRIO", n_chunks_p == 1 ? 2u : 0u));
        uint32_t index_r = env_u32("KG_INDEX_R", 0u);
        if (index_r == 0) index_r = mean * 0.7 >= 640.0 ? 1u : mean * 0.7 >= 320.0 ? 2u : 4u;
        if (index_r != 1 && index_r != 2) index_r = 4;
        while (index_r > 1 && (n_wg % index_r != 0 || kg::kIndexN % index_r != 0)) index_r /= 2;
        // verify workgroups: two per CU.  With eight (until round 3) the pass alone is 15 % faster, but its workgroups take all the
        // registers an ending tag pass frees, and the next tag pass -- the critical chain -- starts behind them: stage 18.3 ->
        // 18.15 ms, 125 Mbp shard 3.18 -> 3.10 (profiles/r03_experiments.md)
        const uint32_t verify_grid = env_u32("KG_VERIFY_GRID", 256u * 2u);
        // The two kernels that usually find nothing to do (no low-complexity block set aside, no overflow group) sit on the
        // stage's critical chain -- in front of every tag pass and behind every verify pass -- and beside the other passes a
        // grid of 2048 / 1024 workgroups takes 0.1 / 0.35 ms just to be scheduled and leave (profiles/r03_kernel_stats.csv);
        // one workgroup per CU leaves in microseconds and is still the whole chip when there is work.
        const uint32_t lowc_grid = env_u32("KG_LOWC_GRID", 256u), ovf_grid = env_u32("KG_OVF_GRID", 256u);
        // per-chunk lists: hits (unordered) and candidates = fingerprint matches (hits + ~0.4 % of the probes) + the
        // ~2 % of the probes whose first tag window decides nothing
        const uint64_t list_slack = (uint64_t)(std::max(std::max(probe_grid, index_grid), verify_grid) + 64) * 4 * kg::kUChunk + 4096;
        uint64_t ucap = ((uint64_t)((double)windows * t->stage_ratio * max_frac) + list_slack + kg::kUChunk - 1) / kg::kUChunk * kg::kUChunk;
        uint64_t ccap = ((uint64_t)((double)windows * (t->stage_ratio * 1.25 + 0.03) * max_frac) + list_slack + kg::kUChunk - 1) /
                        kg::kUChunk * kg::kUChunk;
        if (test_hook("KG_TEST_TINY_LISTS")) ucap = ccap = kg::kUChunk;      // tests: force the resize-and-rerun path
        kg_hit *d_ulist = nullptr;
        uint32_t *d_cused = nullptr, *d_candused = nullptr;
        kg::CandRec *d_cand = nullptr;
        // whatever way this block is left (an error return in the middle of an attempt included), the list blocks go
        // back to the cache with the rest of the scratch once the streams are idle (Scratch's destructor runs later)
        struct ListGuard {
            Scratch &sc;
            void **slot[6];
            ~ListGuard() { for (void **q : slot) if (*q) { sc.adopt(*q); *q = nullptr; } }
        } list_guard{sc, {(void **)&d_ulist, (void **)&d_cused, (void **)&d_cand, (void **)&d_candused, (void **)&d_sortA, (void **)&d_sortB}};
        bool too_skewed = false;
        const uint32_t grab_unit = 256u * (uint32_t)std::max(kg::kProbeN, kg::kIndexN);      // (powers of two: the larger is a multiple of the other)
        const uint32_t probe_grab = (std::max(env_u32("KG_PROBE_GRAB", cap), grab_unit) + grab_unit - 1) / grab_unit * grab_unit;
        uint64_t h_tot[6] = {0, 0, 0, 0, 0, 0};
        HIP_TRY(hipEventRecord(t->ev[1], t->stream));
        for (int attempt = 0; attempt < 3; attempt++) {
            const uint64_t hits_cap = ucap * n_chunks_p;
            const size_t cused_stride = (size_t)(ucap / kg::kUChunk + 1), candused_stride = (size_t)(ccap / kg::kUChunk + 1);
            if ((rc = dalloc(t, (void **)&res->d_hits, hits_cap * sizeof(kg_hit)))) return rc;
            if (progress && (rc = dalloc(t, (void **)&res->d_hit_slots, hits_cap * 4))) return rc;
            if ((rc = dalloc(t, (void **)&d_ulist, ucap * n_chunks_p * sizeof(kg_hit)))) return rc;
            if ((rc = dalloc(t, (void **)&d_cused, cused_stride * n_chunks_p * 4))) return rc;
            if ((rc = dalloc(t, (void **)&d_cand, ccap * n_chunks_p * sizeof(kg::CandRec)))) return rc;
            if ((rc = dalloc(t, (void **)&d_candused, candused_stride * n_chunks_p * 4))) return rc;
            if ((rc = dalloc(t, (void **)&d_sortA, ucap * n_chunks_p * sizeof(kg_hit)))) return rc;
            if ((rc = dalloc(t, (void **)&d_sortB, ucap * n_chunks_p * sizeof(kg_hit)))) return rc;
            {   // one launch for all clears (d_totals: totals, counters and flags of a re-run start over)
                kg::ClearList cl;
                cl.n = 7;
                cl.p[0] = d_cused; cl.words[0] = (uint64_t)cused_stride * n_chunks_p;
                cl.p[1] = d_candused; cl.words[1] = (uint64_t)candused_stride * n_chunks_p;
                cl.p[2] = reinterpret_cast<uint32_t *>(d_pc); cl.words[2] = 48 * 2;
                cl.p[3] = reinterpret_cast<uint32_t *>(d_totals); cl.words[3] = 16;
                cl.p[4] = d_ovfc; cl.words[4] = 8 * kMaxChunks;
                cl.p[5] = d_next; cl.words[5] = (uint64_t)next_stride * n_chunks_p;
                cl.p[6] = d_ghist; cl.words[6] = (uint64_t)groups_stride * n_chunks_p;
                cl.p[7] = nullptr; cl.words[7] = 0;
                uint64_t most = 1;                                      // the grid follows the LARGEST list (the kernel strides)
                for (int k = 0; k < cl.n; k++) most = std::max(most, cl.words[k]);
                most /= 4;
                hipLaunchKernelGGL(kg::clear_many_kernel, dim3((uint32_t)std::min<uint64_t>(4096, (most + 255) / 256 + 1)), dim3(256), 0,
                                   t->stream, cl);
            }
            HIP_TRY(hipEventRecord(t->pev[16], t->stream));               // fork: stream2 starts behind the clears
            HIP_TRY(hipStreamWaitEvent(t->stream2, t->pev[16], 0));
            HIP_TRY(hipStreamWaitEvent(t->stream3, t->pev[16], 0));
            {   // the rows' geometry records (kg_order.hpp): they depend on the batch only, and the verify stream has nothing to do
                // until the first chunk is scattered and probed
                const uint64_t nthr = nblocks * PER;
                hipLaunchKernelGGL((kg::row_geo_kernel<AA>), dim3((uint32_t)((nthr + 255) / 256)), dim3(256), 0, t->stream3, d_blocks,
                                   (uint32_t)nblocks, d_geo);
            }
#define KG_PROBE_ARGS t->d_entries, t->d_tags, t->limit, (uint64_t)t->num_sigs, t->magic
            for (uint32_t c = 0; c < n_chunks_p; c++) {
                const uint32_t lo = (uint32_t)clo[c], nb = (uint32_t)(clo[c + 1] - clo[c]);
                uint64_t *ent_c = d_ent + (uint64_t)c * n_regions_total * cap;
                uint32_t *fill_c = d_fill + (uint64_t)c * n_regions_total;
                uint32_t *next_c = d_next + (size_t)c * next_stride;
                uint32_t *ovfc_c = d_ovfc + 8 * c, *ovf_bucket_c = d_ovf_bucket + (size_t)c * ovf_cap;
                uint64_t *ovf_ent_c = d_ovf_ent + (size_t)c * ovf_cap * kg::kGroup;
                kg_hit *ulist_c = d_ulist + (uint64_t)c * ucap;
                uint32_t *cused_c = d_cused + c * cused_stride, *candused_c = d_candused + c * candused_stride;
                kg::CandRec *cand_c = d_cand + (uint64_t)c * ccap;
                unsigned long long *ucur_c = (unsigned long long *)(d_pc + c), *ccur_c = (unsigned long long *)(d_pc + 8 + c);
                (void)0;
                if (!seq_uploaded && (rc = upload(offsets[cseq[c]], offsets[cseq[c + 1]]))) return rc;
                hipLaunchKernelGGL((kg::part_scatter_kernel<AA>), dim3(n_wg), dim3(kg::kWave * kg::kScatterWaves), lds, t->stream, d_seq,
                                   d_blocks, lo, nb, t->limit, (uint32_t)t->num_sigs, t->m35, part_shift, part_buckets,
                                   cap, ent_c, fill_c, ovfc_c, ovf_cap, ovf_bucket_c, ovf_ent_c, ovfc_c + 1, d_lowc + lo, d_ctr, d_prog, scatter_prio);
                hipStream_t s2 = t->stream2, s3 = t->stream3;
                HIP_TRY(hipEventRecord(t->pev[2 * c], t->stream));
                HIP_TRY(hipStreamWaitEvent(t->stream2, t->pev[2 * c], 0));
                // the low-complexity blocks the scatter pass set aside (usually none: every workgroup reads the count and
                // leaves).  In front of the chunk's tag pass, not behind its scatter pass, and in one-wave workgroups whose
                // 4.9 KB of LDS fit beside a resident scatter workgroup (153 KB of a CU's 160): with four-wave workgroups
                // (15.8 KB) the kernel -- and the tag pass behind it -- waited for the NEXT chunk's scatter pass to leave
                // the CUs (profiles/r02_pipeline.md).
                hipLaunchKernelGGL((kg::lowc_blocks_kernel<AA>), dim3(lowc_grid), dim3(64 * kg::kLowcWaves), 0, s2, d_seq, d_blocks, ovfc_c + 1, d_lowc + lo,
                                   t->limit, (uint32_t)t->num_sigs, t->m35, part_shift, n_wg, cap, ent_c, fill_c, ovfc_c, ovf_cap,
                                   ovf_bucket_c, ovf_ent_c, d_ctr, d_prog);
#define KG_TAG_ARGS t->d_tags, t->limit, (uint64_t)t->num_sigs, ent_c, fill_c, n_wg, cap, part_buckets, part_shift, probe_grab, next_c, cand_c, \
                    candused_c, ccur_c, ccap, d_ctr
#define KG_ULIST_ARGS ulist_c, cused_c, ucur_c, ucap, d_ctr, d_prog
                if (use_bidx) {
#define KG_INDEX_ARGS t->d_bidx, (uint32_t)std::min<uint64_t>(t->tail_start, 0xFFFFFFFFull), ent_c, fill_c, n_wg, cap, \
                      part_buckets, part_shift, probe_grab, next_c, cand_c, candused_c, ccur_c, ccap, d_ctr, index_prio
#define KG_INDEX_LAUNCH(R, X) hipLaunchKernelGGL((kg::bucket_index_kernel<kg::kIndexN, R, X>), dim3(index_grid), dim3(256), 0, s2, KG_INDEX_ARGS)
                    // regions per hand-out by their expected fill (an iteration covers 256 * N / R entry slots of each); the
                    // kernel for tables whose classes are their quotients has no q % 19
                    if (t->bidx_exact) { if (index_r <= 1) KG_INDEX_LAUNCH(1, true); else if (index_r == 2) KG_INDEX_LAUNCH(2, true); else KG_INDEX_LAUNCH(4, true); }
                    else { if (index_r <= 1) KG_INDEX_LAUNCH(1, false); else if (index_r == 2) KG_INDEX_LAUNCH(2, false); else KG_INDEX_LAUNCH(4, false); }
#undef KG_INDEX_LAUNCH
#undef KG_INDEX_ARGS
                }
                else if (counters) hipLaunchKernelGGL((kg::bucket_tag_kernel<true>), dim3(probe_grid), dim3(256), 0, s2, KG_TAG_ARGS, d_prog);
                else hipLaunchKernelGGL((kg::bucket_tag_kernel<false>), dim3(probe_grid), dim3(256), 0, s2, KG_TAG_ARGS, (kg::Progress *)nullptr);
                HIP_TRY(hipEventRecord(t->pev[2 * c + 1], s2));
                HIP_TRY(hipStreamWaitEvent(s3, t->pev[2 * c + 1], 0));
                if (counters) {
                    hipLaunchKernelGGL((kg::verify_kernel<AA, true>), dim3(verify_grid), dim3(256), 0, s3, KG_PROBE_ARGS, cand_c,
                                       candused_c, ccur_c, ccap, KG_ULIST_ARGS, verify_prio);
                    hipLaunchKernelGGL((kg::overflow_probe_kernel<AA, true>), dim3(ovf_grid), dim3(256), 0, s3, KG_PROBE_ARGS,
                                       ovf_bucket_c, ovf_ent_c, ovfc_c, ovf_cap, part_shift, KG_ULIST_ARGS);
                } else {
                    hipLaunchKernelGGL((kg::verify_kernel<AA, false>), dim3(verify_grid), dim3(256), 0, s3, KG_PROBE_ARGS, cand_c,
                                       candused_c, ccur_c, ccap, KG_ULIST_ARGS, verify_prio);
                    hipLaunchKernelGGL((kg::overflow_probe_kernel<AA, false>), dim3(ovf_grid), dim3(256), 0, s3, KG_PROBE_ARGS,
                                       ovf_bucket_c, ovf_ent_c, ovfc_c, ovf_cap, part_shift, KG_ULIST_ARGS);
                }
#undef KG_TAG_ARGS
#undef KG_ULIST_ARGS
                HIP_TRY(hipEventRecord(t->pev[20 + c], s3));                // chunk c verified
                HIP_TRY(hipGetLastError());
                if (c + 1 == n_chunks_p) HIP_TRY(hipEventRecord(t->ev[5], t->stream));   // all chunks scattered
            }
#undef KG_PROBE_ARGS
            seq_uploaded = true;
            // Ordered placement (kg_order.hpp), chunk by chunk, behind the LAST scatter pass and beside the tag passes that are
            // still running: its partition workgroups hold 51 KB of LDS and eight wave slots each, and started beside a scatter
            // pass (105 KB and 16 wave slots of every CU) the two starve each other -- chunk 0's two partition passes took
            // 2.2 + 4.3 ms instead of 0.15 + 0.55 and the scatter pass beside them 7.8 ms instead of 2 (profiles/r03_ordering.md).
            // Beside a tag pass the ordering kernels crawl (a partition pass 1.7-3.9 ms instead of 0.13: every memory access
            // queues behind the tag pass's line gathers) while the tag pass hardly notices them.  KG_ORDER_STREAMS=n (1..4; not
            // the default) gives the chunks' orderings n streams of their own, of the LOWEST priority because that gives them
            // hardware queues of their own (a fourth stream of normal priority shares a queue with the third): the orderings
            // of chunks 0-2 then all crawl beside the last tag passes, single scans 20.1-20.25 ms against 20.4, but twenty
            // scans back to back (bench.py) 21.45 against 21.23 ms per step (profiles/r03_experiments.md).
            const uint32_t n_os = n_chunks_p < 2 ? 0u : std::min(env_u32("KG_ORDER_STREAMS", 0u), 4u);
            const bool early_totals = n_os == 0 && env_u32("KG_EARLY_TOTALS", 1u) != 0;     // (every chunk's ordering on stream3: in order behind every verify pass)
            for (uint32_t k = 0; k < n_os; k++)
                if (!t->ostream[k]) {
                    int pr_least = 0, pr_greatest = 0;
                    HIP_TRY(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
                    HIP_TRY(hipStreamCreateWithPriority(&t->ostream[k], hipStreamNonBlocking, pr_least));
                }
            for (uint32_t c = 0; c < n_chunks_p; c++) {
                const uint32_t lo = (uint32_t)clo[c], nb = (uint32_t)(clo[c + 1] - clo[c]);
                kg_hit *ulist_c = d_ulist + (uint64_t)c * ucap;
                uint32_t *cused_c = d_cused + c * cused_stride;
                unsigned long long *ucur_c = (unsigned long long *)(d_pc + c);
                uint64_t *base_c = d_pc + 16 + c, *ctot_c = d_pc + 32 + c;
                hipStream_t s3 = t->stream;
                if (n_os) {
                    s3 = t->ostream[c % n_os];
                    HIP_TRY(hipStreamWaitEvent(s3, t->ev[5], 0));          // behind the last scatter pass
                }
                HIP_TRY(hipStreamWaitEvent(s3, t->pev[20 + c], 0));
                // group histogram -> group starts -> two partition passes by key range -> ranking inside each group of rows
                {
                    const uint64_t row_lo = (uint64_t)lo * PER, row_hi = row_lo + (uint64_t)nb * PER;
                    const uint32_t g0 = (uint32_t)(row_lo >> gshift);
                    const uint32_t n_groups = nb ? (uint32_t)(((row_hi - 1) >> gshift) - g0 + 1) : 1u;
                    uint32_t *ghist_c = d_ghist + (size_t)c * groups_stride, *gbase_c = d_gbase + (size_t)c * (groups_stride + 1);
                    uint32_t *gcur1_c = d_gcur1 + (size_t)c * (kg::kHDigits + 1), *gcur2_c = d_gcur2 + (size_t)c * groups_stride,
                             *gtile_c = d_gtile + (size_t)c * (kg::kHDigits + 1);
                    kg_hit *sortA_c = d_sortA + (uint64_t)c * ucap, *sortB_c = d_sortB + (uint64_t)c * ucap;
                    const uint32_t ogrid = env_u32("KG_ORDER_GRID", 256u * 3u);
                    hipLaunchKernelGGL(kg::hit_hist_kernel, dim3(ogrid), dim3(kg::kHThreads), (size_t)n_groups * 4, s3, ulist_c, cused_c, ucur_c, ucap,
                                       g0, 6u + gshift, n_groups, ghist_c);
                    hipLaunchKernelGGL(kg::group_scan_kernel, dim3(1), dim3(kg::kGsThreads), 0, s3, ghist_c, n_groups, gbase_c, gcur1_c, gcur2_c, ctot_c, gtile_c);
                    if (n_os && c) HIP_TRY(hipStreamWaitEvent(s3, t->pev[32 + c - 1], 0));      // base of chunk c = base + total of c - 1
                    hipLaunchKernelGGL(kg::chunk_base_kernel, dim3(1), dim3(1), 0, s3, ctot_c, base_c,
                                       c + 1 == n_chunks_p ? d_totals : (uint64_t *)nullptr);
                    if (n_os) HIP_TRY(hipEventRecord(t->pev[32 + c], s3));
                    if (early_totals && c + 1 == n_chunks_p) {
                        // Everything the host wants to know about this attempt is final here -- the list cursors (the last verify
                        // pass is behind us on this stream), the overflow counters, the exact hit total (chunk_base_kernel just
                        // above): it is sent now, and the host reads it, makes the aggregation's allocations and enqueues its
                        // kernels while the last chunk's partition passes and placement still run (the round trip was ~70 us
                        // of every scan, behind the ordering).
                        HIP_TRY(hipMemcpyAsync(t->h_pin, d_pc, 48 * 8, hipMemcpyDeviceToHost, s3));
                        HIP_TRY(hipMemcpyAsync(t->h_pin + 48, d_ovfc, 8 * kMaxChunks * 4, hipMemcpyDeviceToHost, s3));
                        HIP_TRY(hipMemcpyAsync(t->h_pin + 80, d_totals, 48, hipMemcpyDeviceToHost, s3));
                        HIP_TRY(hipEventRecord(t->pev[19], s3));
                    }
                    hipLaunchKernelGGL((kg::hit_partition_kernel<true>), dim3(ogrid), dim3(kg::kHThreads), 0, s3, ulist_c, cused_c, ucur_c, ucap,
                                       gbase_c, n_groups, g0, 6u + gshift, gcur1_c, sortA_c, ucap, gtile_c);
                    hipLaunchKernelGGL((kg::hit_partition_kernel<false>), dim3(ogrid), dim3(kg::kHThreads), 0, s3, sortA_c, cused_c, ucur_c, ucap,
                                       gbase_c, n_groups, g0, 6u + gshift, gcur2_c, sortB_c, ucap, gtile_c);
                    const bool place_staged = gshift == 10 && env_u32("KG_PLACE_STAGED", 1u) != 0;
                    hipLaunchKernelGGL((kg::group_place_kernel<AA>), dim3(std::min(n_groups, 256u * 8u)), dim3(kg::kHThreads),
                                       kg::group_place_lds(gshift, place_staged), s3,
                                       sortB_c, gbase_c, n_groups, g0, gshift, (uint32_t)row_lo, (uint32_t)row_hi, d_geo, (uint64_t)n_rows,
                                       place_staged ? 1u : 0u, base_c, res->d_hits, hits_cap, d_offs, res->d_hit_slots);
                }
                HIP_TRY(hipGetLastError());
            }

            for (uint32_t k = 0; k < n_os; k++) {
                HIP_TRY(hipEventRecord(t->pev[40 + k], t->ostream[k]));
                HIP_TRY(hipStreamWaitEvent(t->stream, t->pev[40 + k], 0));
            }
            HIP_TRY(hipEventRecord(t->pev[17], t->stream2));              // join
            HIP_TRY(hipStreamWaitEvent(t->stream, t->pev[17], 0));
            HIP_TRY(hipEventRecord(t->pev[18], t->stream3));
            HIP_TRY(hipStreamWaitEvent(t->stream, t->pev[18], 0));
            HIP_TRY(hipEventRecord(t->ev[7], t->stream));
            st.scan_launches++;
            const uint64_t *h_pc = t->h_pin;
            const uint32_t *h_ovf = reinterpret_cast<const uint32_t *>(t->h_pin + 48);
            static_assert(8 * kMaxChunks * 4 <= 32 * 8, "overflow counters must fit their pinned words");
            HIP_TRY(hipEventRecord(t->ev[2], t->stream));                 // end of the scan stage (of this attempt)
            if (early_totals) {
                HIP_TRY(hipEventSynchronize(t->pev[19]));                 // (the ordering of the last chunk may still be running)
            } else {
                HIP_TRY(hipMemcpyAsync(t->h_pin, d_pc, 48 * 8, hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipMemcpyAsync(t->h_pin + 48, d_ovfc, 8 * kMaxChunks * 4, hipMemcpyDeviceToHost, t->stream));
                HIP_TRY(hipMemcpyAsync(t->h_pin + 80, d_totals, 48, hipMemcpyDeviceToHost, t->stream));   // pinned: one host round trip for all three
                HIP_TRY(hipStreamSynchronize(t->stream));
            }
            for (int k = 0; k < 6; k++) h_tot[k] = t->h_pin[80 + k];
            uint64_t need_u = 0, need_c = 0;
            uint32_t max_ovf = 0, guard = 0;
            for (uint32_t c = 0; c < n_chunks_p; c++) {
                need_u = std::max(need_u, h_pc[c]); need_c = std::max(need_c, h_pc[8 + c]);
                max_ovf = std::max(max_ovf, h_ovf[8 * c]);
                guard |= h_ovf[8 * c + 2];
            }
            if (getenv("KG_DEBUG"))
                fprintf(stderr, "[kg] partition attempt %d: %u chunks (largest %llu of %llu blocks), overflow groups <= %u (cap %u), hit list <= %llu "
                                "(cap %llu), candidates <= %llu (cap %llu), regions/chunk %llu x %u entries, %u buckets, shift %u, %u scatter "
                                "workgroups, hits %llu, %s\n",
                        attempt, n_chunks_p, (unsigned long long)max_chunk, (unsigned long long)nblocks, max_ovf, ovf_cap,
                        (unsigned long long)need_u, (unsigned long long)ucap, (unsigned long long)need_c, (unsigned long long)ccap,
                        (unsigned long long)n_regions_total, cap, part_buckets, part_shift, n_wg, (unsigned long long)h_pc[16 + n_chunks_p], use_bidx ? "byte home index" : "tags");
            const bool redo = guard || max_ovf > ovf_cap || need_u > ucap || need_c > ccap;
            if (redo && early_totals) HIP_TRY(hipStreamSynchronize(t->stream));   // the attempt is thrown away: its last kernels first
            if (guard) { too_skewed = true; st.fallback = 2; break; }    // the scatter pass's spin guard fired: direct path
            if (max_ovf > ovf_cap) { too_skewed = true; st.fallback = 1; break; }   // more overflow than provisioned: direct path
            n_hits = h_pc[16 + n_chunks_p];
            if (need_u <= ucap && need_c <= ccap) break;
            // a list was too small: now the exact need is known (masks are cleared and everything is redone)
            dfree(t, d_ulist); dfree(t, d_cused); dfree(t, d_cand); dfree(t, d_candused); dfree(t, res->d_hits);   // both streams are idle
            dfree(t, d_sortA); dfree(t, d_sortB);                         // (the ordering buffers are sized by ucap as well)
            dfree(t, res->d_hit_slots); res->d_hit_slots = nullptr;
            d_ulist = nullptr; d_cused = nullptr; d_cand = nullptr; d_candused = nullptr; res->d_hits = nullptr;
            d_sortA = nullptr; d_sortB = nullptr;
            if (attempt == 2) return fail(KG_ERR_DEVICE, "hit list overflow after resize (internal error)");
            // which wave fills which reservation chunk differs from run to run: one partly used chunk per wave on top
            if (need_c > ccap) { ccap = (need_c + list_slack + kg::kUChunk - 1) / kg::kUChunk * kg::kUChunk; ucap = std::max(ucap, ccap); }   // hits <= candidates
            else ucap = (need_u + list_slack + kg::kUChunk - 1) / kg::kUChunk * kg::kUChunk;
        }
        if (too_skewed) {
            dfree(t, res->d_hits); dfree(t, res->d_hit_slots);
            res->d_hits = nullptr; res->d_hit_slots = nullptr;
        } else {
            st.windows_valid = counters ? (int64_t)h_tot[2] : -1;
            st.slots_inspected = counters ? (int64_t)h_tot[3] : -1;
            st.lookup_ran_off = h_tot[5] ? 1 : 0;
            if (windows) {
                double ratio = (double)n_hits / (double)windows * 1.1 + 1e-3;
                if (ratio > t->stage_ratio) t->stage_ratio = ratio > 1.0 ? 1.0 : ratio;
            }
            st.n_hits = (int64_t)n_hits;
            part_done = true;
            st.partitioned = 1;
            st.part_chunks = (int32_t)n_chunks_p; st.part_buckets = (int32_t)part_buckets; st.part_shift = (int32_t)part_shift;
            st.part_levels = use_bidx ? 4 : 1;
        }
    } while (0);
    if (!part_done) {
        st.scan_launches = 0;
        if (!seq_uploaded) { if ((rc = upload(offsets[0], offsets[n_seqs]))) return rc; seq_uploaded = true; }
    // ---- scan: encode + probe + staged compaction; re-run once if the staging area was too small ----
    // persistent grid: enough workgroups to fill 256 CUs, few enough that per-wave staging chunks stay small
    const uint32_t scan_grid = env_u32("KG_SCAN_GRID", 256u * 8u);
    const uint32_t stage_chunk = env_u32("KG_STAGE_CHUNK", 256u);
    // the table's bit-per-slot digest as the direct kernel's first question (tables of <= kHbitsMaxSlots slots; not for scans
    // that count the slots they inspect): config 5's scan 2.28 -> 1.80 ms (r04 c34)
    // KG_DIRECT_FILTER: 0 never, 1 (default) when the tags no longer fit an XCD's 4 MB L2 (below that the bit is one more
    // dependent load in front of an L2 hit), 2 whenever the table has the digest (tests)
    const uint32_t filter_mode = env_u32("KG_DIRECT_FILTER", 1u);
    const uint32_t *d_hbits_scan = (counters || filter_mode == 0 || (filter_mode == 1 && t->limit <= (4ull << 20))) ? nullptr : t->d_hbits;
    // rows probed together per lane: three; six behind the digest, where two probes out of three end at the bit (1.80 -> 1.75 ms)
    uint32_t rpg = AA ? 1u : env_u32("KG_SCAN_RPG", d_hbits_scan ? 6u : 3u);
    if (rpg != 1 && rpg != 2 && rpg != 3 && rpg != 6) rpg = 3;
    uint64_t stage_cap = (uint64_t)((double)windows * t->stage_ratio) + 4096 +
                         (uint64_t)scan_grid * kg::kWavesPerWG * stage_chunk;
    if (stage_cap > 0xFFFFFF00ull) stage_cap = 0xFFFFFF00ull;
    if (test_hook("KG_TEST_TINY_LISTS")) stage_cap = 256;                    // tests: force the resize-and-rerun path
    kg_hit *d_stage = nullptr;
    uint32_t *d_stage_slot = nullptr;                                        // KG_F_PROGRESS: the found slots, parallel to d_stage
    struct StageGuard { Scratch &sc; kg_hit *&p; uint32_t *&q; ~StageGuard() { if (p) sc.adopt(p); if (q) sc.adopt(q); } } stage_guard{sc, d_stage, d_stage_slot};
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = dalloc(t, (void **)&d_stage, stage_cap * sizeof(kg_hit)))) return rc;
        if (progress && (rc = dalloc(t, (void **)&d_stage_slot, stage_cap * 4))) return rc;
        unsigned long long *d_cursor = (unsigned long long *)(d_totals + 1);
        unsigned long long *d_ctr = (unsigned long long *)(d_totals + 2);
        HIP_TRY(hipMemsetAsync(d_totals, 0, 64, t->stream));
        HIP_TRY(hipEventRecord(t->ev[1], t->stream));
        if (nblocks) {
            uint64_t wgs = (nblocks + kg::kWavesPerWG - 1) / kg::kWavesPerWG;
            uint32_t grid = (uint32_t)(wgs < scan_grid ? wgs : scan_grid);      // persistent waves stride over the blocks
#define KG_SCAN_ARGS t->d_entries, t->d_tags, t->limit, (uint64_t)t->num_sigs, t->magic, t->m35, d_seq, d_blocks, (uint32_t)nblocks, \
                     d_counts, d_bsb, d_stage, d_cursor, stage_cap, stage_chunk, d_ctr, d_prog, d_stage_slot, \
                     d_hbits_scan, t->tail_start
#define KG_SCAN_LAUNCH(C, R) hipLaunchKernelGGL((kg::scan_kernel<AA, C, R>), dim3(grid), dim3(kg::kWave * kg::kWavesPerWG), 0, \
                                                t->stream, KG_SCAN_ARGS)
            if (AA) {
                if (counters) KG_SCAN_LAUNCH(true, 1); else KG_SCAN_LAUNCH(false, 1);
            } else {
                constexpr int R1 = AA ? 1 : 1, R2 = AA ? 1 : 2, R3 = AA ? 1 : 3, R6 = AA ? 1 : 6;
                if (counters) {
                    if (rpg == 1) KG_SCAN_LAUNCH(true, R1); else if (rpg == 2) KG_SCAN_LAUNCH(true, R2);
                    else if (rpg == 3) KG_SCAN_LAUNCH(true, R3); else KG_SCAN_LAUNCH(true, R6);
                } else {
                    if (rpg == 1) KG_SCAN_LAUNCH(false, R1); else if (rpg == 2) KG_SCAN_LAUNCH(false, R2);
                    else if (rpg == 3) KG_SCAN_LAUNCH(false, R3); else KG_SCAN_LAUNCH(false, R6);
                }
            }
#undef KG_SCAN_LAUNCH
#undef KG_SCAN_ARGS
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(t->ev[2], t->stream));
        st.scan_launches++;
        if ((rc = prefix_sum(t, d_counts, n_rows, d_offs, d_partial, d_totals))) return rc;
        uint64_t h_tot[6] = {0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(t->h_pin + 80, d_totals, 48, hipMemcpyDeviceToHost, t->stream));    // (pinned words: no staging copy on the host)
        HIP_TRY(hipStreamSynchronize(t->stream));
        for (int k = 0; k < 6; k++) h_tot[k] = t->h_pin[80 + k];
        n_hits = n_rows ? h_tot[0] : 0;
        st.windows_valid = counters ? (int64_t)h_tot[2] : -1;
        st.slots_inspected = counters ? (int64_t)h_tot[3] : -1;
        st.lookup_ran_off = h_tot[5] ? 1 : 0;
        if (h_tot[1] <= stage_cap) break;
        // staging overflow: now the exact need is known
        dfree(t, d_stage); d_stage = nullptr;
        dfree(t, d_stage_slot); d_stage_slot = nullptr;
        if (attempt == 1) return fail(KG_ERR_DEVICE, "staging overflow after resize (internal error)");
        stage_cap = h_tot[1];
    }
    if (windows) {
        double ratio = (double)n_hits / (double)windows * 1.1 + 1e-3;
        if (ratio > t->stage_ratio) t->stage_ratio = ratio > 1.0 ? 1.0 : ratio;
    }
    st.n_hits = (int64_t)n_hits;

    // ---- ordered placement ----
    if ((rc = dalloc(t, (void **)&res->d_hits, n_hits * sizeof(kg_hit)))) return rc;
    if (progress && (rc = dalloc(t, (void **)&res->d_hit_slots, (n_hits ? n_hits : 1) * 4))) return rc;
    if (nblocks) {
        uint32_t grid = (uint32_t)((nblocks + kg::kWavesPerWG - 1) / kg::kWavesPerWG);
        hipLaunchKernelGGL((kg::place_kernel<AA>), dim3(grid), dim3(kg::kWave * kg::kWavesPerWG), 0, t->stream, d_blocks,
                           (uint32_t)nblocks, d_counts, d_offs, d_bsb, rpg, d_stage, res->d_hits, d_stage_slot, res->d_hit_slots);
    }
    }
    if (progress) {
        // kmersFound / found-so-far: the distinct slots of the hit records (a bitmap over the stream's slots)
        uint32_t *d_bitmap = nullptr;
        const uint64_t n_words = (t->limit + 31) / 32 + 1;
        if ((rc = sc.get(&d_bitmap, (size_t)n_words))) return rc;
        HIP_TRY(hipMemsetAsync(d_bitmap, 0, n_words * 4, t->stream));
        if (n_hits)
            hipLaunchKernelGGL(kg::mark_found_kernel, dim3((uint32_t)std::min<uint64_t>(2048, (n_hits + 255) / 256)), dim3(256), 0, t->stream,
                               res->d_hit_slots, n_hits, d_bitmap);
        hipLaunchKernelGGL(kg::count_found_kernel, dim3((uint32_t)std::min<uint64_t>(2048, (n_words + 255) / 256)), dim3(256), 0, t->stream,
                           d_bitmap, n_words, d_prog);
    }
    hipLaunchKernelGGL((kg::container_starts_kernel<AA>), dim3((uint32_t)((n_cont + 1 + 255) / 256)), dim3(256), 0, t->stream,
                       d_ibase, (uint32_t)n_seqs, d_offs, n_rows, d_totals, res->d_chs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(t->ev[3], t->stream));

    // ---- aggregation: CALL records and OTU votes ----
    const bool aggregate = !(p->flags & KG_F_SKIP_AGGREGATE);
    if (aggregate && (rc = aggregate_stage(t, p, res, sc, n_seqs, n_cont, n_hits, PER, d_partial, d_totals, nullptr, longest < (1ll << 30))))
        return rc;
    HIP_TRY(hipEventRecord(t->ev[4], t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    st.n_calls = aggregate ? (int64_t)t->h_pin[kPinCalls] : 0;
    if (progress) {
        kg::Progress h;
        HIP_TRY(hipMemcpy(&h, d_prog, sizeof h, hipMemcpyDeviceToHost));
        kg_progress &g = res->progress;
        for (int f = 0; f <= 10; f++) g.first_visited[f] = h.first[f] == ~0ull ? -1 : (int64_t)h.first[f];
        g.last_visited = (int64_t)h.last_plus1 - 1;
        g.first_beyond = h.first_beyond == ~0ull ? -1 : (int64_t)h.first_beyond;
        g.walk_ran_off = h.walk_ran_off ? 1 : 0;
        g.stream_slots = (int64_t)t->limit;
        for (int f = 0; f <= 10; f++) g.found_upto[f] = g.first_visited[f] < 0 ? 0 : (int64_t)h.found_upto[f];
        g.kmers_found = (int64_t)h.kmers_found;
        res->has_progress = true;
    }
    st.agg_pieces = aggregate ? (int32_t)std::min<uint64_t>(t->h_pin[kPinPieces], 0x7FFFFFFF) : 0;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[1], t->ev[2])); st.ms_scan = ms;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[2], t->ev[3])); st.ms_order = ms;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[3], t->ev[4])); st.ms_aggregate = ms;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[0], t->ev[4])); st.ms_total = ms;
    if (st.partitioned) {
        // the passes of different chunks overlap: "scatter" = until the last chunk is scattered, "tail" = what is left
        // of the tag / verify passes after that; ms_part_tag is kept for layout compatibility
        HIP_TRY(hipEventElapsedTime(&ms, t->ev[1], t->ev[5])); st.ms_part_scatter = ms;
        st.ms_part_tag = 0;
        HIP_TRY(hipEventElapsedTime(&ms, t->ev[5], t->ev[7])); st.ms_part_verify = ms;
    }
    return KG_OK;
}

int scan_entry(kg_table *t, const kg_params *p, const uint8_t *seq, bool on_device, const int64_t *offsets, int64_t n_seqs,
               kg_result **out)
{
    if (!t || !p || !offsets || !out || n_seqs < 0) return fail(KG_ERR_ARG, "null or negative argument");
    if (n_seqs > 0x7FFFFFF0ll / 6) return fail(KG_ERR_LIMIT, "too many sequences in one batch");
    if (p->min_hits < 2)
        return fail(KG_ERR_UNSUPPORTED, "minHits < 2: the reference throws in processSetOfHits (KGJ:442); refusing");
    // One scan at a time per table: the streams, events, pinned counter words and the block cache's "freed when the
    // stream is idle" rule are per table.  A second thread is turned away instead of corrupting them.
    if (t->busy.exchange(1) != 0)
        return fail(KG_ERR_BUSY, "another kg_scan* is in flight on this kg_table (one scan at a time per table; open a second table "
                                 "object for concurrent scans)");
    struct BusyGuard { kg_table *t; ~BusyGuard() { t->busy.store(0); } } busy_guard{t};
    t->fail_alloc_at = test_hook("KG_TEST_FAIL_ALLOC");
    t->alloc_count = 0;
    HIP_TRY(hipSetDevice(t->device));
    int64_t total = offsets[n_seqs] - offsets[0];
    if (total < 0) return fail(KG_ERR_ARG, "offsets must be non-decreasing");
    if (!seq && total > 0) return fail(KG_ERR_ARG, "null sequence buffer");
    kg_result *r = new (std::nothrow) kg_result();
    if (!r) return fail(KG_ERR_NOMEM, "out of host memory");
    r->tab = t;
    uint8_t *d_seq = nullptr;
    int rc = KG_OK;
    if (!on_device) {
        size_t end = (size_t)offsets[n_seqs];
        rc = dalloc(t, (void **)&d_seq, end + 16);       // filled by scan_impl (upload overlapped with the scan where possible)
    }
    if (rc == KG_OK) {
        const uint8_t *s = on_device ? seq : d_seq;
        const uint8_t *h = on_device ? nullptr : seq;
        rc = p->aa ? scan_impl<true>(t, p, s, h, offsets, n_seqs, r) : scan_impl<false>(t, p, s, h, offsets, n_seqs, r);
    }
    (void)hipStreamSynchronize(t->stream);
    if (d_seq) dfree(t, d_seq);
    if (rc != KG_OK) {
        std::string keep = g_err;
        kg_result_free(r);
        g_err = keep;
        return rc;
    }
    *out = r;
    return KG_OK;
}

template <typename T>
const T *host_view(kg_result *r, void *&slot, const T *d, size_t n)
{
    if (slot) return (const T *)slot;
    if (!d && n) { g_err = "record kind not computed (KG_F_SKIP_AGGREGATE?)"; return nullptr; }
    if (hipSetDevice(r->tab->device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    void *h = nullptr;
    if (r->tab->pins.get(&h, n ? n * sizeof(T) : 64) != hipSuccess) { g_err = "pinned host allocation failed"; return nullptr; }
    if (n && hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) {
        r->tab->pins.put(h);
        g_err = "device to host copy failed";
        return nullptr;
    }
    slot = h;
    return (const T *)h;
}

}  // namespace

extern "C" {

int kg_scan(kg_table *t, const kg_params *p, const uint8_t *seq, const int64_t *offsets, int64_t n_seqs, kg_result **out)
{
    return scan_entry(t, p, seq, false, offsets, n_seqs, out);
}

int kg_scan_device(kg_table *t, const kg_params *p, const uint8_t *d_seq, const int64_t *offsets, int64_t n_seqs,
                   kg_result **out)
{
    return scan_entry(t, p, d_seq, true, offsets, n_seqs, out);
}

int kg_aggregate_hits(int device, const kg_params *p, const kg_hit *hits, const int64_t *container_hit_start, int64_t n_seqs,
                      const kg_otu *otu_init, kg_result **out)
{
    if (!p || !container_hit_start || !out || n_seqs < 0) return fail(KG_ERR_ARG, "null or negative argument");
    if (n_seqs > 0x7FFFFFF0ll / 6) return fail(KG_ERR_LIMIT, "too many sequences in one batch");
    if (p->min_hits < 2)
        return fail(KG_ERR_UNSUPPORTED, "minHits < 2: the reference throws in processSetOfHits (KGJ:442); refusing");
    const uint32_t PER = p->aa ? 1u : 6u;
    const uint64_t n_cont = (uint64_t)n_seqs * PER;
    if (container_hit_start[0] != 0) return fail(KG_ERR_ARG, "container_hit_start[0] must be 0");
    for (uint64_t c = 0; c < n_cont; c++)
        if (container_hit_start[c + 1] < container_hit_start[c]) return fail(KG_ERR_ARG, "container_hit_start must be non-decreasing");
    const uint64_t n_hits = (uint64_t)container_hit_start[n_cont];
    if (n_hits && !hits) return fail(KG_ERR_ARG, "null hit records");
    if (n_hits > 0xFFFFFF00ull) return fail(KG_ERR_LIMIT, "more than 2^32-256 hit records");
    kg_table *t = nullptr;
    int rc = table_new(device, &t);
    if (rc) return rc;
    kg_result *r = new (std::nothrow) kg_result();
    if (!r) { kg_table_close(t); return fail(KG_ERR_NOMEM, "out of host memory"); }
    r->tab = t; r->own_tab = true; r->per = PER;
    rc = [&]() -> int {                 // (the scratch blocks go back to the context's cache before the context can be closed)
        Scratch sc(t);
        int rc2;
        uint64_t *d_partial = nullptr, *d_totals = nullptr;
        kg_otu *d_init = nullptr;
        if ((rc2 = dalloc(t, (void **)&r->d_hits, (n_hits ? n_hits : 1) * sizeof(kg_hit)))) return rc2;
        if ((rc2 = dalloc(t, (void **)&r->d_chs, (n_cont + 1) * 8))) return rc2;
        if ((rc2 = sc.get(&d_partial, (size_t)(n_cont / kg::kScanChunk + 2)))) return rc2;
        if ((rc2 = sc.get(&d_totals, 8))) return rc2;
        if (otu_init && n_seqs && (rc2 = sc.get(&d_init, (size_t)n_seqs))) return rc2;
        HIP_TRY(hipMemsetAsync(d_totals, 0, 64, t->stream));
        if (n_hits) HIP_TRY(hipMemcpyAsync(r->d_hits, hits, n_hits * sizeof(kg_hit), hipMemcpyHostToDevice, t->stream));
        HIP_TRY(hipMemcpyAsync(r->d_chs, container_hit_start, (n_cont + 1) * 8, hipMemcpyHostToDevice, t->stream));
        if (d_init) HIP_TRY(hipMemcpyAsync(d_init, otu_init, (size_t)n_seqs * sizeof(kg_otu), hipMemcpyHostToDevice, t->stream));
        // (caller-supplied records: positions are whatever the caller says, so long containers stay in one piece)
        if ((rc2 = aggregate_stage(t, p, r, sc, n_seqs, n_cont, n_hits, PER, d_partial, d_totals, d_init, false))) return rc2;
        HIP_TRY(hipStreamSynchronize(t->stream));
        r->st.n_seqs = n_seqs; r->st.n_containers = (int64_t)n_cont; r->st.n_hits = (int64_t)n_hits;
        r->st.n_calls = (int64_t)t->h_pin[kPinCalls];
        r->st.windows_valid = -1; r->st.slots_inspected = -1;
        return KG_OK;
    }();
    if (rc != KG_OK) { std::string keep = g_err; kg_result_free(r); g_err = keep; return rc; }
    *out = r;
    return KG_OK;
}

int kg_process_set_of_hits(int device, const kg_params *p, const kg_hit *hits, int32_t n_hits, int32_t current_fi, kg_otu *otu,
                           kg_call *call, int32_t *called, int32_t *new_current_fi, int32_t *keeps_last_two)
{
    if (!p || !hits || !otu || !call || !called || !new_current_fi || !keeps_last_two) return fail(KG_ERR_ARG, "null argument");
    if (n_hits < 2)
        return fail(KG_ERR_UNSUPPORTED, "processSetOfHits on fewer than two hits: the reference throws (hits.get(numHits-2), KGJ:442); refusing");
    if (otu->n < 0 || otu->n > KG_OI_BUFSZ) return fail(KG_ERR_ARG, "oICounts holds more than OI_BUFSZ entries");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(KG_ERR_DEVICE, "no HIP device: libkmerguts_hip needs an MI355X (gfx950) GPU; there is no CPU path");
    if (device < 0 || device >= ndev) return fail(KG_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    kg_hit *d_hits = nullptr;
    uint8_t *d_small = nullptr;                        // kg_otu | kg_call | int32 x 4
    const size_t small = sizeof(kg_otu) + sizeof(kg_call) + 16;
    HIP_TRY(hipMalloc((void **)&d_hits, (size_t)n_hits * sizeof(kg_hit)));
    hipError_t e = hipMalloc((void **)&d_small, small);
    if (e == hipSuccess) e = hipMemcpy(d_hits, hits, (size_t)n_hits * sizeof(kg_hit), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_small, 0, small);
    if (e == hipSuccess) e = hipMemcpy(d_small, otu, sizeof(kg_otu), hipMemcpyHostToDevice);
    uint8_t h_small[sizeof(kg_otu) + sizeof(kg_call) + 16];
    if (e == hipSuccess) {
        kg::AggParams ap;
        ap.min_hits = p->min_hits; ap.min_weighted_hits = p->min_weighted_hits;
        ap.max_gap = p->max_gap; ap.order_constraint = p->order_constraint ? 1 : 0;
        hipLaunchKernelGGL(kg::process_set_single_kernel, dim3(1), dim3(64), 0, nullptr, d_hits, n_hits, current_fi, ap,
                           (kg_otu *)d_small, (kg_call *)(d_small + sizeof(kg_otu)), (int32_t *)(d_small + sizeof(kg_otu) + sizeof(kg_call)));
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(h_small, d_small, small, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_hits);
    if (d_small) (void)hipFree(d_small);
    if (e != hipSuccess) return fail(KG_ERR_DEVICE, std::string("processSetOfHits on the device failed: ") + hipGetErrorString(e));
    memcpy(otu, h_small, sizeof(kg_otu));
    memcpy(call, h_small + sizeof(kg_otu), sizeof(kg_call));
    int32_t o3[4];
    memcpy(o3, h_small + sizeof(kg_otu) + sizeof(kg_call), 16);
    *called = o3[0]; *new_current_fi = o3[1]; *keeps_last_two = o3[2];
    return KG_OK;
}

int kg_result_stats(const kg_result *r, kg_stats *out)
{
    if (!r || !out) return fail(KG_ERR_ARG, "null argument");
    *out = r->st;
    return KG_OK;
}

const kg_hit *kg_result_hits(kg_result *r)
{
    return r ? host_view(r, r->h_hits, r->d_hits, (size_t)r->st.n_hits) : nullptr;
}
const int64_t *kg_result_container_hit_start(kg_result *r)
{
    return r ? host_view(r, r->h_chs, r->d_chs, (size_t)r->st.n_containers + 1) : nullptr;
}
const kg_call *kg_result_calls(kg_result *r)
{
    return r ? host_view(r, r->h_calls, r->d_calls, (size_t)r->st.n_calls) : nullptr;
}
const int64_t *kg_result_container_call_start(kg_result *r)
{
    if (!r) return nullptr;
    if (!r->d_ccs) { g_err = "calls not computed (KG_F_SKIP_AGGREGATE)"; return nullptr; }
    return host_view(r, r->h_ccs, r->d_ccs, (size_t)r->st.n_containers + 1);
}
const kg_otu *kg_result_otu(kg_result *r)
{
    if (!r) return nullptr;
    if (!r->d_otu) { g_err = "OTU votes not computed (KG_F_SKIP_AGGREGATE)"; return nullptr; }
    return host_view(r, r->h_otu, r->d_otu, (size_t)r->st.n_seqs);
}
const uint8_t *kg_result_hit_events(kg_result *r)
{
    if (!r) return nullptr;
    if (!r->d_ev) { g_err = "events not computed (KG_F_SKIP_AGGREGATE)"; return nullptr; }
    return host_view(r, r->h_ev, r->d_ev, (size_t)r->st.n_hits);
}
const uint8_t *kg_result_container_tail_events(kg_result *r)
{
    if (!r) return nullptr;
    if (!r->d_tail_ev) { g_err = "events not computed (KG_F_SKIP_AGGREGATE)"; return nullptr; }
    return host_view(r, r->h_tail_ev, r->d_tail_ev, (size_t)r->st.n_containers);
}
const uint32_t *kg_result_hit_slots(kg_result *r)
{
    if (!r || !r->has_progress) { g_err = "hit slots are recorded by KG_F_PROGRESS scans only"; return nullptr; }
    return host_view<uint32_t>(r, r->h_hit_slots, r->d_hit_slots, (size_t)r->st.n_hits);
}

int kg_result_progress(const kg_result *r, kg_progress *out)
{
    if (!r || !out) return fail(KG_ERR_ARG, "null argument");
    if (!r->has_progress) return fail(KG_ERR_ARG, "not a KG_F_PROGRESS scan");
    *out = r->progress;
    return KG_OK;
}

int kg_result_copy_hits(kg_result *r, int64_t first, int64_t count, kg_hit *dst)
{
    if (!r || first < 0 || count < 0 || first + count > r->st.n_hits) return fail(KG_ERR_ARG, "hit range out of bounds");
    if (count == 0) return KG_OK;
    if (!dst) return fail(KG_ERR_ARG, "null destination");
    HIP_TRY(hipSetDevice(r->tab->device));
    // pageable destinations go through the table's two cached pinned blocks, 64 MiB at a time: the device-to-host copy
    // of piece k+1 runs while piece k is moved into the caller's memory
    hipPointerAttribute_t attr;
    const bool pinned_dst = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (pinned_dst) {
        HIP_TRY(hipMemcpy(dst, r->d_hits + first, (size_t)count * sizeof(kg_hit), hipMemcpyDeviceToHost));
        return KG_OK;
    }
    const size_t piece = (64u << 20) / sizeof(kg_hit);
    void *stage[2] = {nullptr, nullptr};
    for (auto &st : stage)
        if (r->tab->pins.get(&st, piece * sizeof(kg_hit)) != hipSuccess) {
            if (stage[0]) r->tab->pins.put(stage[0]);
            return fail(KG_ERR_NOMEM, "pinned staging allocation failed");
        }
    hipStream_t s = r->tab->stream;
    hipEvent_t done[2] = {r->tab->ev[6], r->tab->ev[0]};      // idle outside a scan
    int rc = KG_OK;
    int64_t sent = 0, got = 0;
    int which = 0;
    auto issue = [&](int w) {
        const int64_t n = std::min<int64_t>((int64_t)piece, count - sent);
        hipError_t e = hipMemcpyAsync(stage[w], r->d_hits + first + sent, (size_t)n * sizeof(kg_hit), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(done[w], s);
        if (e != hipSuccess) rc = fail(KG_ERR_DEVICE, std::string("device to host copy failed: ") + hipGetErrorString(e));
        sent += n;
    };
    issue(0);
    while (rc == KG_OK && got < count) {
        if (sent < count) issue(which ^ 1);
        if (rc != KG_OK) break;
        if (hipEventSynchronize(done[which]) != hipSuccess) { rc = fail(KG_ERR_DEVICE, "device to host copy failed"); break; }
        const int64_t n = std::min<int64_t>((int64_t)piece, count - got);
        memcpy(dst + got, stage[which], (size_t)n * sizeof(kg_hit));
        got += n;
        which ^= 1;
    }
    (void)hipStreamSynchronize(s);
    r->tab->pins.put(stage[0]); r->tab->pins.put(stage[1]);
    return rc;
}

int64_t kg_table_live_device_bytes(kg_table *t)
{
    return t ? (int64_t)t->cache.live_bytes() : 0;
}

int kg_restore_hits_device(int device, const kg_hit *d_src, int64_t n_hits, const int64_t *d_seq_first, int64_t n_seqs,
                           const int64_t *d_dst_first, const int32_t *d_container_shift, kg_hit *d_dst, void *stream)
{
    if (n_hits < 0 || n_seqs < 0) return fail(KG_ERR_ARG, "negative count");
    if (n_hits == 0) return KG_OK;
    if (!d_src || !d_seq_first || !d_dst_first || !d_container_shift || !d_dst || n_seqs == 0) return fail(KG_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    const uint32_t grid = (uint32_t)std::min<int64_t>((n_hits + 1023) / 1024, 256 * 16);
    hipLaunchKernelGGL(kg::restore_hits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_src, (uint64_t)n_hits, d_seq_first,
                       (uint64_t)n_seqs, d_dst_first, d_container_shift, d_dst);
    HIP_TRY(hipGetLastError());
    return KG_OK;
}

const void *kg_result_device_hits(const kg_result *r) { return r ? r->d_hits : nullptr; }
const void *kg_result_device_calls(const kg_result *r) { return r ? r->d_calls : nullptr; }
const void *kg_result_device_otu(const kg_result *r) { return r ? r->d_otu : nullptr; }
const void *kg_result_device_container_hit_start(const kg_result *r) { return r ? r->d_chs : nullptr; }
const void *kg_result_device_container_call_start(const kg_result *r) { return r ? r->d_ccs : nullptr; }

}  // extern "C"
