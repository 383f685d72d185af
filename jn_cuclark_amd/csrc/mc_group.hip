// mc_group.hip -- several GPUs behind one handle (include/mc_group.h): replicas or shards, the
// device-to-device row exchange, the k-way merge on the owner.  Replaces the multi-device half of the
// reference's device manager (src/CuClarkDB.cu:118-215, :516-559, :842-851, :909-928).
//
// Stream discipline (shards).  Member m works on its context's stream `batch & 1`: wait until the owners
// have read the rows of the batch that used this slot before -> H2D -> query kernel (sparse rows of ALL
// reads) -> event.  Owner j then, on its own stream: wait for every member's event -> pull its read range
// of their rows (hipMemcpyPeerAsync: xGMI, point to point, every link carries 1/G of the rows once) ->
// k-way merge + top-2 -> D2H of its range -> event `done[j]`.  Two batches are in flight per device, so
// the exchange and merge of one overlap the query kernel of the next.
#include "../../include/mc_group.h"
#include "mc_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

using mcint::fail;

namespace {

struct GSlot {
    uint32_t *d_ptr = nullptr;
    uint16_t *d_con = nullptr;
    uint16_t *d_rows = nullptr;      // shards: rows of all reads; replicas: rows of the batch (extended)
    uint16_t *d_recv = nullptr;      // shards: (W-1) x per x row_len, the other members' rows of my read range
    uint16_t *d_final = nullptr;
    uint16_t *d_merged = nullptr;    // shards + extended: merged rows of my read range
    hipEvent_t ev_query = nullptr;
};

struct GBatch {
    uint32_t *h_ptr = nullptr;
    uint16_t *h_con = nullptr;
    uint16_t *h_final = nullptr;
    uint16_t *h_rows = nullptr;
    std::vector<hipEvent_t> done;    // per member
    bool submitted = false;
    uint32_t first = 0, count = 0;   // the members that worked on the batch as it was last submitted
};

} // namespace

struct mc_group {
    std::vector<mc_ctx *> ctx;
    uint32_t k = 0, maxhits = 0, num_targets = 0;
    uint64_t htsize = 0;
    mc_group_info info{};
    bool loaded = false;
    std::vector<GBatch> batches;
    std::vector<GSlot> slots;        // [member * 2 + slot]
    uint8_t *h_slab = nullptr;       // the pinned buffers of all batches
    std::vector<uint8_t *> d_slab;   // per member: the device buffers of its two slots
    uint64_t max_reads = 0, max_con = 0, per = 0;
    bool want_rows = false;
    // batches may be submitted in any order (the host packs them on several threads): slots and, for
    // replicas, members are dealt by SUBMISSION order
    uint64_t n_submitted = 0;
    // shards: S parts of the table x G groups that each hold all of it.  Member m = part m % S of group m / S;
    // members past S * G (N not a multiple of S) hold nothing.  A batch goes to ONE group.
    uint32_t S = 1, G = 1;
    std::vector<int> last_on_slot;   // [group * 2 + slot]: the batch whose rows the owners last pulled from this slot
    // a table larger than all members together (the reference's swapDbParts, CuClarkDB.cu:775-815): cycles * N parts, the
    // members hold those of one cycle; what a change of cycle needs to read the files again
    uint32_t cycles = 1, cycle = 0;
    std::string base;
    int key_bytes = 0, cycle_kind = 0;      // index of the parts (1 minimizer lines, 2 super-k-mer records): one for all cycles
    uint32_t sampling = 1;
    double cycle_fill = 0.0;
};

namespace {

uint32_t W(const mc_group *g) { return (uint32_t)g->ctx.size(); }

// read range of owner j (part j of the batch's group)
void range_of(const mc_group *g, uint64_t n_reads, uint32_t j, uint64_t &lo, uint64_t &cnt)
{
    const uint64_t per = (n_reads + g->S - 1) / g->S;
    lo = std::min<uint64_t>(n_reads, (uint64_t)j * per);
    cnt = std::min<uint64_t>(n_reads, lo + per) - lo;
}

int free_batches(mc_group *g)
{
    for (uint32_t m = 0; m < W(g); m++) {
        (void)hipSetDevice(g->ctx[m]->device);
        (void)hipDeviceSynchronize();
    }
    if (g->h_slab) (void)hipHostFree(g->h_slab);
    g->h_slab = nullptr;
    for (auto &b : g->batches) {
        for (size_t m = 0; m < b.done.size(); m++)
            if (b.done[m]) { (void)hipSetDevice(g->ctx[m]->device); (void)hipEventDestroy(b.done[m]); }
    }
    g->batches.clear();
    g->n_submitted = 0; g->last_on_slot.assign((size_t)g->G * 2, -1);
    for (size_t i = 0; i < g->slots.size(); i++) {
        GSlot &s = g->slots[i];
        (void)hipSetDevice(g->ctx[i / 2]->device);
        if (s.ev_query) (void)hipEventDestroy(s.ev_query);
    }
    g->slots.clear();
    for (size_t m = 0; m < g->d_slab.size(); m++) {
        for (auto &cs : g->ctx[m]->slots) cs = mcint::Slot();           // lent in replica mode: the slab owns them
        if (g->d_slab[m]) { (void)hipSetDevice(g->ctx[m]->device); (void)hipFree(g->d_slab[m]); }
    }
    g->d_slab.clear();
    return MC_OK;
}

// HBM a full replica of the table needs (the choice between replicas and shards)
uint64_t bytes_for_table(const mc_ctx *c, uint64_t n_keys, uint64_t nb)
{
    if (mcint::minimizer_index_possible(c, n_keys)) {
        double per_line = 10.0;         // a dense fill the loader would still accept (slower, but no exchange between devices)
        if (const char *e = getenv("MC_MZ_FILL")) { const double v = atof(e); if (v >= 1.0 && v <= 64.0) per_line = v; }
        return mcint::index_bytes(n_keys, 1, per_line);
    }
    return nb * 128 + n_keys * 14;        // bucket lines (worst case 128 B) next to the raw arrays during the fill
}

} // namespace

extern "C" {

int mc_group_open(mc_group **out, const int *devices, uint32_t n_devices, uint32_t k, uint64_t htsize,
                  uint32_t num_targets, uint32_t maxhits)
{
    if (!out) return fail(MC_EINVAL, "out is NULL");
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) return fail(MC_ENODEVICE, "no HIP device visible");
    std::vector<int> devs;
    const char *env = devices ? nullptr : getenv("MC_GROUP_DEVICES");
    if (env && *env) {
        // explicit member list, e.g. 0,0,1 (a device may repeat: the sharded path rehearsed on one card)
        for (const char *p = env; *p;) {
            char *end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v < 0 || v >= visible) return fail(MC_ENODEVICE, "MC_GROUP_DEVICES names a device that is not visible");
            devs.push_back((int)v);
            p = *end ? end + 1 : end;
        }
        if (devs.empty()) return fail(MC_EINVAL, "MC_GROUP_DEVICES is not a list of device numbers");
    } else if (devices) {
        if (n_devices < 1) return fail(MC_EINVAL, "empty device list");
        devs.assign(devices, devices + n_devices);
    } else {
        const uint32_t n = n_devices ? n_devices : (uint32_t)visible;          // 0 = all (CuClarkDB.cu:146-150)
        if (n > (uint32_t)visible)
            return fail(MC_ENODEVICE, std::to_string(n) + " devices requested. Insufficient devices found. Abort.");
        for (uint32_t i = 0; i < n; i++) devs.push_back((int)i);
    }
    if (devs.size() > 16) return fail(MC_EINVAL, "at most 16 members");
    mc_group *g = new mc_group();
    g->k = k; g->htsize = htsize; g->num_targets = num_targets; g->maxhits = maxhits;
    for (int d : devs) {
        mc_ctx *c = nullptr;
        const int rc = mc_open(&c, d, k, htsize, num_targets, maxhits);
        if (rc != MC_OK) { mc_group_close(g); return rc; }
        g->ctx.push_back(c);
    }
    // peer access between every pair of distinct devices (CuClarkDB.cu:201-213); without it
    // hipMemcpyPeerAsync still works (staged), only slower
    std::set<int> distinct(devs.begin(), devs.end());
    bool all_peer = true;
    for (int a : distinct) {
        for (int b : distinct) {
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) { all_peer = false; continue; }
            (void)hipSetDevice(a);
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) all_peer = false;
            (void)hipGetLastError();
        }
    }
    g->info.n_members = (uint32_t)devs.size();
    g->info.peer_access = all_peer ? 1u : 0u;
    *out = g;
    return MC_OK;
}

int mc_group_close(mc_group *g)
{
    if (!g) return MC_OK;
    free_batches(g);
    for (mc_ctx *c : g->ctx) mc_close(c);
    delete g;
    return MC_OK;
}

namespace {
// the parts [cycle * N, cycle * N + N) of the cycles * N the table is cut into, one per member
int load_cycle(mc_group *g, mcint::DbFileStream &F, uint32_t cycle)
{
    const uint32_t n = W(g);
    std::vector<int> saved(n);
    for (uint32_t m = 0; m < n; m++) { saved[m] = g->ctx[m]->index_mode; if (g->cycle_kind) g->ctx[m]->index_mode = g->cycle_kind; }
    const int rc = mcint::load_streamed(g->ctx.data(), n, F, g->cycles * n, cycle * n, g->cycle_fill);
    for (uint32_t m = 0; m < n; m++) g->ctx[m]->index_mode = saved[m];
    if (rc == MC_OK) {
        g->cycle = cycle; g->info.cycle = cycle;
        if (!g->cycle_kind) g->cycle_kind = (int)g->ctx[0]->info.index_kind;
    }
    return rc;
}
} // namespace

int mc_group_set_cycle(mc_group *g, uint32_t cycle)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    if (!g->loaded) return fail(MC_ESTATE, "mc_group_set_cycle before a database was loaded");
    if (cycle >= g->cycles) return fail(MC_EINVAL, "no such cycle");
    if (cycle == g->cycle) return MC_OK;
    int rc = mc_group_sync(g);
    if (rc) return rc;
    mcint::DbFileStream F;
    if ((rc = F.open(g->base.c_str(), g->key_bytes, g->sampling, g->htsize, 0, g->htsize)) != MC_OK) return rc;
    g->loaded = false;
    if ((rc = load_cycle(g, F, cycle)) != MC_OK) return rc;
    g->loaded = true;
    return MC_OK;
}

int mc_group_load_db(mc_group *g, const char *base, int key_bytes, uint32_t sampling, int mode)
{
    if (!g || !base) return fail(MC_EINVAL, "group/base is NULL");
    if (key_bytes != 2 && key_bytes != 4 && key_bytes != 8) return fail(MC_EINVAL, "key_bytes must be 2, 4 or 8");
    if (mode < MC_GROUP_AUTO || mode > MC_GROUP_SHARDS) return fail(MC_EINVAL, "bad mode");
    g->loaded = false;
    mcint::DbFileStream F;
    int rc = F.open(base, key_bytes, sampling, g->htsize, 0, g->htsize);
    if (rc) return rc;
    const uint32_t n = W(g);
    mc_ctx *c0 = g->ctx[0];
    const bool mz = mcint::minimizer_index_possible(c0, F.n_keys_kept);

    // the budget (CuClarkDB.cu:516-559): free HBM of the smallest device, shared by the members on it
    uint64_t free_min = ~0ull;
    for (uint32_t m = 0; m < n; m++) {
        size_t fr = 0, tot = 0;
        if ((rc = mcint::set_dev(g->ctx[m])) != MC_OK) return rc;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        uint32_t sharing = 0;
        for (uint32_t o = 0; o < n; o++) sharing += g->ctx[o]->device == g->ctx[m]->device ? 1u : 0u;
        free_min = std::min<uint64_t>(free_min, fr / sharing);
    }
    if (const char *e = getenv("MC_GROUP_HBM_BYTES")) { const uint64_t v = strtoull(e, nullptr, 10); if (v) free_min = std::min(free_min, v); }
    const uint64_t need_one = bytes_for_table(c0, F.n_keys_kept, g->htsize);
    const bool want_auto = mode == MC_GROUP_AUTO && !getenv("MC_GROUP_MODE");
    if (mode == MC_GROUP_AUTO) {
        if (const char *e = getenv("MC_GROUP_MODE")) {
            if (!strcmp(e, "replicas")) mode = MC_GROUP_REPLICAS;
            else if (!strcmp(e, "shards")) mode = MC_GROUP_SHARDS;
        }
    }
    const bool by_lines = mz && !(getenv("MC_GROUP_SHARD") && !strcmp(getenv("MC_GROUP_SHARD"), "buckets"));
    // How many parts.  The reference cuts the table into as few parts as the devices' memory dictates
    // (minParts, CuClarkDB.cu:529-559) -- every further part repeats the front half of the kernel for every read.
    // Here: S = the smallest part count whose share fits a device at an acceptable fill, G = N / S groups that
    // each hold the whole table and take every G-th batch; S = 1 is "replicas".  Bucket-range shards (no
    // minimizer index) and a forced MC_GROUP_SHARDS keep one group of N parts.
    uint32_t s_min = 0;
    if (by_lines) s_min = mcint::min_parts(F.n_keys_kept, n, free_min, MC_GROUP_MAX_FILL);
    else if (need_one + (4ull << 30) <= free_min) s_min = 1;
    if (mode == MC_GROUP_AUTO) mode = (n == 1 || s_min == 1) ? MC_GROUP_REPLICAS : MC_GROUP_SHARDS;
    if (n == 1) mode = MC_GROUP_REPLICAS;
    uint32_t S = 1;
    if (mode == MC_GROUP_SHARDS) {
        S = n;
        if (want_auto && by_lines && s_min >= 2) S = n / (n / s_min);        // G = n / s_min groups, spread evenly
        if (const char *e = getenv("MC_GROUP_PARTS")) { const long v = atol(e); if (v >= 2 && v <= (long)n && by_lines) S = (uint32_t)v; }
    }

    uint32_t shard_kind = 0;
    for (;;) {
        const uint32_t G = mode == MC_GROUP_SHARDS ? n / S : 1, n_act = mode == MC_GROUP_SHARDS ? S * G : n;
        shard_kind = 0;
        if (mode == MC_GROUP_REPLICAS) {
            if (mz) rc = mcint::load_streamed(g->ctx.data(), n, F, 1, 0, 0.0);
            if (!mz || (rc == MC_ENOMEM && !want_auto)) {
                for (uint32_t m = 0; m < n; m++) {           // bucket-line tables (or the fallback to them), device by device
                    rc = mc_load_db(g->ctx[m], base, key_bytes, sampling, 0, 0);
                    if (rc != MC_OK) break;
                }
            }
        } else if (by_lines) {
            shard_kind = 1;
            rc = mcint::load_streamed(g->ctx.data(), n_act, F, S, 0, 0.0);
        } else {
            shard_kind = 2;                                   // the reference's own partition (CuClarkDB.cu:552-559)
            for (uint32_t m = 0; m < n; m++) {
                rc = mc_load_db(g->ctx[m], base, key_bytes, sampling, g->htsize * m / n, g->htsize * (m + 1) / n);
                if (rc != MC_OK) break;
            }
        }
        // AUTO and the estimate was too kind (a table just under one card's HBM, a table whose lines overflow more
        // than the estimate assumes): cut it into more parts instead of giving up
        if (rc == MC_ENOMEM && want_auto && by_lines && S < n) {
            mode = MC_GROUP_SHARDS;
            uint32_t next = S + 1;
            while (next < n && n / next == n / S) next++;                      // the next S that changes the group count
            S = n / (n / next);
            fprintf(stderr, "libmcclark: %s; cutting the table into %u parts\n", mc_last_error(), S);
            continue;
        }
        g->S = mode == MC_GROUP_SHARDS ? S : 1; g->G = mode == MC_GROUP_SHARDS ? G : n;
        break;
    }
    if (rc == MC_ENOMEM && mz && !getenv("MC_GROUP_NO_LINES_FALLBACK")) {      // (the variable: how the tests get past this to the database cycles)
        // Last resort, whatever the mode asked for: the bucket-line table (about 3x slower to query, a third of the
        // memory at dense fills) -- whole on every member when it fits one, else cut by the reference's bucket ranges
        // (CuClarkDB.cu:552-559).  Said aloud; mc_db_info of the members carries index_fallback = 1.
        fprintf(stderr, "libmcclark: %s; falling back to the bucket-line table\n", mc_last_error());
        std::vector<int> saved(n);
        for (uint32_t m = 0; m < n; m++) { saved[m] = g->ctx[m]->index_mode; g->ctx[m]->index_mode = 0; }
        bool whole = mode == MC_GROUP_REPLICAS || n == 1;
        if (whole) {
            for (uint32_t m = 0; m < n; m++) {
                rc = mc_load_db(g->ctx[m], base, key_bytes, sampling, 0, 0);
                if (rc != MC_OK) break;
            }
            if (rc == MC_OK) { mode = MC_GROUP_REPLICAS; shard_kind = 0; g->S = 1; g->G = n; }
        }
        if (n > 1 && (!whole || rc == MC_ENOMEM)) {
            for (uint32_t m = 0; m < n; m++) {
                rc = mc_load_db(g->ctx[m], base, key_bytes, sampling, g->htsize * m / n, g->htsize * (m + 1) / n);
                if (rc != MC_OK) break;
            }
            if (rc == MC_OK) { mode = MC_GROUP_SHARDS; shard_kind = 2; g->S = n; g->G = 1; }
        }
        for (uint32_t m = 0; m < n; m++) {
            g->ctx[m]->index_mode = saved[m];
            if (rc == MC_OK) g->ctx[m]->info.index_fallback = 1u;
        }
    }
    uint32_t forced_cycles = 0;
    if (const char *e = getenv("MC_GROUP_CYCLES")) { const long v = atol(e); if (v >= 2 && v <= 64) forced_cycles = (uint32_t)v; }
    g->cycles = 1; g->cycle = 0; g->cycle_kind = 0; g->cycle_fill = 0.0;
    if ((rc == MC_ENOMEM || (forced_cycles && rc == MC_OK)) && mz && by_lines && n <= mc::MERGE_MAX_SRCS - 1) {
        // Larger than all members together (the reference's swapDbParts, CuClarkDB.cu:775-815): C * N parts, N at a time.  The
        // caller classifies every batch once per cycle and the rows add up (mc_group_set_cycle, MC_F_FOLLOWUP).
        const uint32_t s_all = mcint::min_parts(F.n_keys_kept, 4096, free_min, MC_GROUP_MAX_FILL);
        uint32_t C = forced_cycles ? forced_cycles : std::max<uint32_t>(2, s_all ? (s_all + n - 1) / n : 2);
        for (;; C++) {
            g->cycles = C; g->cycle_kind = 0;
            g->cycle_fill = mcint::choose_fill(F.n_keys_kept, C * n, free_min);
            rc = load_cycle(g, F, 0);
            if (rc != MC_ENOMEM || forced_cycles || C >= 64) break;
            fprintf(stderr, "libmcclark: %s; %u cycles\n", mc_last_error(), C + 1);
        }
        if (rc == MC_OK) {
            fprintf(stderr, "libmcclark: the database does not fit the %u device(s) together: %u parts, %u at a time (%u cycles per file)\n", n, C * n, n, C);
            mode = MC_GROUP_SHARDS; shard_kind = 1; g->S = n; g->G = 1;
            for (mc_ctx *c : g->ctx) c->info.index_fallback = 0u;
        } else {
            g->cycles = 1;
        }
    }
    if (rc == MC_ENOMEM)
        return fail(MC_ENOMEM, std::string("the database does not fit the ") + std::to_string(n) + " device(s) of the group (" +
                                   mc_last_error() + "); use more devices (-d)");
    if (rc != MC_OK) return rc;
    g->info.mode = (uint32_t)mode;
    g->info.shard_kind = shard_kind;
    g->info.n_keys = F.n_keys_kept;
    g->info.bytes_needed_one = need_one;
    g->info.bytes_free_min = free_min;
    g->info.device_bytes_max = 0;
    g->info.n_shards = g->S; g->info.n_groups = g->G;
    g->info.n_cycles = g->cycles; g->info.cycle = g->cycle;
    g->base = base; g->key_bytes = key_bytes; g->sampling = sampling;
    for (mc_ctx *c : g->ctx) if (c->db_loaded) g->info.device_bytes_max = std::max<uint64_t>(g->info.device_bytes_max, c->info.device_bytes);
    g->last_on_slot.assign((size_t)g->G * 2, -1);
    g->loaded = true;
    return MC_OK;
}

int mc_group_get_info(mc_group *g, mc_group_info *out)
{
    if (!g || !out) return fail(MC_EINVAL, "NULL argument");
    *out = g->info;
    return MC_OK;
}

int mc_group_member(mc_group *g, uint32_t i, mc_ctx **out)
{
    if (!g || !out || i >= W(g)) return fail(MC_EINVAL, "bad member index");
    *out = g->ctx[i];
    return MC_OK;
}

int mc_group_alloc_batches(mc_group *g, uint32_t n_batches, uint64_t max_reads, uint64_t max_con, int want_rows)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    if (!g->loaded) return fail(MC_ESTATE, "mc_group_alloc_batches before a database was loaded");
    if (n_batches < 1 || max_reads < 1) return fail(MC_EINVAL, "n_batches and max_reads must be >= 1");
    if (max_con > 0xFFFFFFFFull) return fail(MC_EINVAL, "max_containers exceeds the 32-bit offsets of the batch format");
    free_batches(g);
    if (max_con < 8) max_con = 8;
    max_con = (max_con + 7) & ~7ull;
    const uint32_t n = W(g);
    const bool shards = g->info.mode == MC_GROUP_SHARDS;
    g->max_reads = max_reads; g->max_con = max_con; g->want_rows = want_rows != 0;
    g->per = (max_reads + g->S - 1) / g->S;
    const size_t row_len = 2 * (size_t)g->maxhits + 2;
    auto oom = [&](const char *what, hipError_t e) {
        free_batches(g);
        return fail(MC_ENOMEM, std::string(what) + ": " + hipGetErrorString(e) + " -- use more, smaller batches (-b)");
    };
    int rc = mcint::set_dev(g->ctx[0]); if (rc) return rc;
    // ONE pinned allocation for the buffers of all batches and ONE device allocation per member: pinning and
    // mapping cost per call, and 4 x n_batches calls were a third of a 2 M-read run (38 ms).
    auto up = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
    const size_t b_ptr = up((max_reads + 1) * 4), b_con = up(max_con * 2), b_fin = up(max_reads * MC_FINAL_ROW * 2),
                 b_rows = g->want_rows ? up(max_reads * row_len * 2) : 0;
    const size_t per_batch = b_ptr + b_con + b_fin + b_rows;
    // portable: every device of the group copies from / into these
    hipError_t e = hipHostMalloc((void **)&g->h_slab, per_batch * n_batches, hipHostMallocPortable);
    if (e != hipSuccess) return oom("pinned batch buffers", e);
    g->batches.resize(n_batches);
    for (uint32_t i = 0; i < n_batches; i++) {
        GBatch &b = g->batches[i];
        uint8_t *base = g->h_slab + per_batch * i;
        b.h_ptr = (uint32_t *)base;
        b.h_con = (uint16_t *)(base + b_ptr);
        b.h_final = (uint16_t *)(base + b_ptr + b_con);
        b.h_rows = g->want_rows ? (uint16_t *)(base + b_ptr + b_con + b_fin) : nullptr;
        b.done.assign(n, nullptr);
        for (uint32_t m = 0; m < n; m++) {
            if ((rc = mcint::set_dev(g->ctx[m])) != MC_OK) { free_batches(g); return rc; }
            e = hipEventCreateWithFlags(&b.done[m], hipEventDisableTiming);
            if (e != hipSuccess) return oom("events", e);
        }
    }
    g->slots.resize((size_t)n * 2);
    g->d_slab.assign(n, nullptr);
    const size_t d_ptr = up((max_reads + 1) * 4), d_con = up(max_con * 2);
    const size_t d_rows = shards ? up(max_reads * row_len * 2) : (g->want_rows ? up(max_reads * row_len * 2) : 0);
    const size_t d_recv = shards ? up((size_t)(g->cycles > 1 ? g->S : (g->S > 1 ? g->S - 1 : 1)) * g->per * row_len * 2) : 0;
    const size_t d_fin = up((shards ? g->per : max_reads) * MC_FINAL_ROW * 2);
    const size_t d_mrg = shards && g->want_rows ? up(g->per * row_len * 2) : 0;
    const size_t per_slot = d_ptr + d_con + d_rows + d_recv + d_fin + d_mrg;
    for (uint32_t m = 0; m < n; m++) {
        if (shards && m >= g->S * g->G) break;            // a member without a part of the table
        if ((rc = mcint::set_dev(g->ctx[m])) != MC_OK) { free_batches(g); return rc; }
        e = hipMalloc((void **)&g->d_slab[m], per_slot * 2);
        if (e != hipSuccess) return oom("device batch buffers", e);
        for (int si = 0; si < 2; si++) {
            GSlot &s = g->slots[(size_t)m * 2 + si];
            uint8_t *base = g->d_slab[m] + per_slot * si;
            s.d_ptr = (uint32_t *)base;
            s.d_con = (uint16_t *)(base + d_ptr);
            s.d_rows = d_rows ? (uint16_t *)(base + d_ptr + d_con) : nullptr;
            s.d_recv = d_recv ? (uint16_t *)(base + d_ptr + d_con + d_rows) : nullptr;
            s.d_final = (uint16_t *)(base + d_ptr + d_con + d_rows + d_recv);
            s.d_merged = d_mrg ? (uint16_t *)(base + d_ptr + d_con + d_rows + d_recv + d_fin) : nullptr;
            e = hipEventCreateWithFlags(&s.ev_query, hipEventDisableTiming);
            if (e != hipSuccess) return oom("events", e);
            if (!shards) {          // replicas run through the member's own batch queues: lend it the buffers
                mcint::Slot &cs = g->ctx[m]->slots[si];
                cs.d_ptr = s.d_ptr; cs.d_con = s.d_con; cs.d_final = s.d_final; cs.d_rows = s.d_rows;
            }
        }
    }
    return MC_OK;
}

int mc_group_batch_buffers(mc_group *g, uint32_t batch, uint32_t **reads_ptr, uint16_t **containers,
                           uint16_t **final_rows, uint16_t **sparse_rows)
{
    if (!g || batch >= g->batches.size()) return fail(MC_EINVAL, "bad batch index");
    GBatch &b = g->batches[batch];
    if (reads_ptr) *reads_ptr = b.h_ptr;
    if (containers) *containers = b.h_con;
    if (final_rows) *final_rows = b.h_final;
    if (sparse_rows) *sparse_rows = b.h_rows;
    return MC_OK;
}

int mc_group_submit(mc_group *g, uint32_t batch, uint64_t n_reads, uint64_t n_con, uint32_t flags)
{
    if (!g || batch >= g->batches.size()) return fail(MC_EINVAL, "bad batch index");
    if (n_reads > g->max_reads || n_con > g->max_con) return fail(MC_EINVAL, "batch larger than allocated");
    if (!(flags & (MC_F_FINAL | MC_F_ROWS))) return fail(MC_EINVAL, "flags select no output");
    if ((flags & MC_F_ROWS) && !g->want_rows) return fail(MC_ESTATE, "sparse rows were not allocated");
    if ((flags & MC_F_FOLLOWUP) && (g->cycles < 2 || !g->want_rows))
        return fail(MC_ESTATE, "MC_F_FOLLOWUP needs a database that is loaded in cycles, and batches with sparse rows");
    GBatch &b = g->batches[batch];
    if (n_reads && b.h_ptr[n_reads] != n_con) return fail(MC_EINVAL, "reads_ptr[n_reads] != n_containers");
    const uint32_t n = W(g);
    const size_t row_len = 2 * (size_t)g->maxhits + 2;
    int rc;

    if (!g->loaded) return fail(MC_ESTATE, "mc_group_submit before a database was loaded");
    if (g->info.mode != MC_GROUP_SHARDS) {
        // replicas: the batch goes to one member, through that context's three queues (copy in, compute, copy out)
        const uint32_t m = (uint32_t)(g->n_submitted++ % n);
        const int si = 0;
        mc_ctx *c = g->ctx[m];
        hipStream_t st = c->s_out;
        if ((rc = mcint::set_dev(c)) != MC_OK) return rc;
        rc = mcint::submit_batch(c, b.h_ptr, b.h_con, b.h_final, b.h_rows, n_reads, n_con, flags, b.done[m]);
        if (rc) return rc;
        (void)st; (void)si;
        b.first = m; b.count = 1;
        b.submitted = true;
        return MC_OK;
    }

    // ---- shards ----------------------------------------------------------------------------------
    // the batch goes to ONE group (round-robin over the G groups that each hold the whole table); inside the group
    // every member sees it
    const uint32_t S = g->S, gi = (uint32_t)(g->n_submitted % g->G), m0 = gi * S;
    const int si = (int)((g->n_submitted / g->G) & 1u);
    g->n_submitted++;
    const int prev = g->last_on_slot[(size_t)gi * 2 + si];
    g->last_on_slot[(size_t)gi * 2 + si] = (int)batch;
    // every member of the group: (slot free?) -> H2D -> rows of all reads for its share of the table
    for (uint32_t p = 0; p < S; p++) {
        mc_ctx *c = g->ctx[m0 + p];
        GSlot &s = g->slots[(size_t)(m0 + p) * 2 + si];
        hipStream_t st = c->streams[si];
        if ((rc = mcint::set_dev(c)) != MC_OK) return rc;
        if (prev >= 0 && prev != (int)batch)                        // the owners pulled from this slot's rows
            for (uint32_t j = 0; j < S; j++)
                if (j != p) HIPCHK(hipStreamWaitEvent(st, g->batches[prev].done[m0 + j], 0));
        if (n_reads) {
            HIPCHK(hipMemcpyAsync(s.d_ptr, b.h_ptr, (n_reads + 1) * 4, hipMemcpyHostToDevice, st));
            if (n_con) HIPCHK(hipMemcpyAsync(s.d_con, b.h_con, n_con * 2, hipMemcpyHostToDevice, st));
            rc = mcint::launch_query(c, s.d_ptr, s.d_con, n_reads, n_con, MC_F_ROWS, nullptr, s.d_rows, st);
            if (rc) return rc;
        }
        HIPCHK(hipEventRecord(s.ev_query, st));
    }
    // every owner: pull its read range from the others, merge, top-2, back to the host
    for (uint32_t j = 0; j < S; j++) {
        mc_ctx *c = g->ctx[m0 + j];
        GSlot &s = g->slots[(size_t)(m0 + j) * 2 + si];
        hipStream_t st = c->streams[si];
        if ((rc = mcint::set_dev(c)) != MC_OK) return rc;
        uint64_t lo, cnt;
        range_of(g, n_reads, j, lo, cnt);
        if (cnt) {
            const uint16_t *srcs[mc::MERGE_MAX_SRCS];
            uint32_t r = 0;
            for (uint32_t i = 0; i < S; i++) {
                GSlot &o = g->slots[(size_t)(m0 + i) * 2 + si];
                if (i == j) { srcs[i] = s.d_rows + lo * row_len; continue; }
                HIPCHK(hipStreamWaitEvent(st, o.ev_query, 0));
                uint16_t *dst = s.d_recv + (size_t)r * g->per * row_len;
                HIPCHK(hipMemcpyPeerAsync(dst, c->device, o.d_rows + lo * row_len, g->ctx[m0 + i]->device, cnt * row_len * 2, st));
                srcs[i] = dst;
                r++;
            }
            uint32_t n_srcs = S;
            if (flags & MC_F_FOLLOWUP) {      // the rows the earlier cycles gave these reads: one more source (CuClarkDB.cu:932-948)
                uint16_t *dst = s.d_recv + (size_t)r * g->per * row_len;
                HIPCHK(hipMemcpyAsync(dst, b.h_rows + lo * row_len, cnt * row_len * 2, hipMemcpyHostToDevice, st));
                srcs[n_srcs++] = dst;
            }
            rc = mcint::launch_merge_result(c, srcs, n_srcs, cnt, (flags & MC_F_ROWS) ? s.d_merged : nullptr,
                                            (flags & MC_F_FINAL) ? s.d_final : nullptr, st);
            if (rc) return rc;
            if (flags & MC_F_FINAL)
                HIPCHK(hipMemcpyAsync(b.h_final + lo * MC_FINAL_ROW, s.d_final, cnt * MC_FINAL_ROW * 2, hipMemcpyDeviceToHost, st));
            if (flags & MC_F_ROWS)
                HIPCHK(hipMemcpyAsync(b.h_rows + lo * row_len, s.d_merged, cnt * row_len * 2, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipEventRecord(b.done[m0 + j], st));
    }
    b.first = m0; b.count = S;
    b.submitted = true;
    return MC_OK;
}

int mc_group_wait(mc_group *g, uint32_t batch)
{
    if (!g || batch >= g->batches.size()) return fail(MC_EINVAL, "bad batch index");
    GBatch &b = g->batches[batch];
    if (!b.submitted) return fail(MC_ESTATE, "batch was never submitted");
    for (uint32_t m = b.first; m < b.first + b.count; m++) {       // the members that worked on it
        int rc = mcint::set_dev(g->ctx[m]); if (rc) return rc;
        HIPCHK(hipEventSynchronize(b.done[m]));
    }
    return MC_OK;
}

int mc_group_sync(mc_group *g)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    for (mc_ctx *c : g->ctx) { int rc = mc_sync(c); if (rc) return rc; }
    return MC_OK;
}

int mc_group_free_batches(mc_group *g)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    for (mc_ctx *c : g->ctx) (void)mc_text_free(c);
    return free_batches(g);
}

/* FASTQ text batches (mc_text_*, csrc/mc_ingest.hip) for a group whose members each hold the whole table: buffer b lives on
 * member b % N (its buffer b / N), so the batches of a file go round the members.  A table that is cut into parts has no
 * text path (every part would have to cut and pack every batch): MC_ESTATE, the caller packs on the host. */
int mc_group_text_alloc(mc_group *g, uint32_t n_buffers, uint64_t max_text, uint64_t max_reads, uint64_t max_con)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    if (!g->loaded) return fail(MC_ESTATE, "mc_group_text_alloc before a database was loaded");
    if (g->info.mode != MC_GROUP_REPLICAS) return fail(MC_ESTATE, "text batches need a group whose members hold the whole table");
    const uint32_t n = W(g);
    for (uint32_t m = 0; m < n; m++) {
        const uint32_t mine = (n_buffers + n - 1 - m) / n;
        if (!mine) continue;
        const int rc = mc_text_alloc(g->ctx[m], mine, max_text, max_reads, max_con);
        if (rc != MC_OK) { for (mc_ctx *c : g->ctx) (void)mc_text_free(c); return rc; }
    }
    return MC_OK;
}
int mc_group_text_buffers(mc_group *g, uint32_t buffer, uint32_t **hdr, uint32_t **len, uint16_t **fin)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    return mc_text_buffers(g->ctx[buffer % W(g)], buffer / W(g), hdr, len, fin);
}
int mc_group_text_submit(mc_group *g, uint32_t buffer, const uint8_t *text, uint64_t n_bytes)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    return mc_text_submit(g->ctx[buffer % W(g)], buffer / W(g), text, n_bytes);
}
int mc_group_text_wait(mc_group *g, uint32_t buffer, uint64_t *n_reads, uint32_t *status)
{
    if (!g) return fail(MC_EINVAL, "group is NULL");
    return mc_text_wait(g->ctx[buffer % W(g)], buffer / W(g), n_reads, status);
}

} // extern "C"
