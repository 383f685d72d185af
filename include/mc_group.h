/*
 * mc_group.h -- several GPUs behind ONE handle, in one process (part of libmcclark.so).
 *
 * The reference's CuClarkDB object drives all devices itself: it discovers them and enables peer
 * access (src/CuClarkDB.cu:118-215), cuts the table into per-device bucket ranges sized by free memory
 * (:516-559), sends EVERY read batch to EVERY device (:842-851), pulls the partial sparse rows to device 0
 * through a blocking cudaMemcpyPeer binary tree with a mergeKernel per hop (:909-928), runs resultKernel
 * there (:963-968), and cycles table parts through the devices when even all of them together are too small
 * (swapDbParts, :775-815).  `-d <n>` selects how many devices (src/main.cc:43-69; 0 = all,
 * CuClarkDB.cu:146-150).  mc_group is that object for MI355X:
 *
 *   replicas   the table fits one GPU (288 GB: every table the reference targets, up to ~12e9 k-mers):
 *              every GPU holds all of it, batches are dealt round-robin, no exchange at all;
 *   shards     it does not: the table is cut into S parts -- as FEW as the cards' memory dictates, like the
 *              reference's minParts (CuClarkDB.cu:529-559), because every part repeats the front half of the kernel
 *              for every read -- and the N GPUs form G = N / S groups that each hold the whole table and take
 *              every G-th batch (S = 1 is "replicas").  Inside a group GPU p holds part p of the minimizer index
 *              (mc_load_db_part; bucket ranges as in the reference when that index is not available: then one
 *              group of N), every GPU of the group gets the batch, each produces sparse rows for all its reads; rows are exchanged device-to-device as a
 *              reduce-scatter by read range (GPU j receives every GPU's rows for read range j:
 *              hipMemcpyPeerAsync over xGMI, all links busy once, nothing funnels into device 0), merged
 *              by one k-way kernel and top-2 on the owner, which copies its range of final rows to the
 *              host.  Integer rows: the result is bit-identical to the unsharded run for any device count.
 *
 * The host sees the same batch interface as with one context (mc_api.h): pinned input and result
 * buffers owned by the library, submit, wait.  Plain C ABI; errors via mc_last_error().
 */
#ifndef MC_GROUP_H
#define MC_GROUP_H

#include "mc_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_group mc_group;

#define MC_GROUP_AUTO     0   /* replicas if the table fits every device, else shards */
#define MC_GROUP_REPLICAS 1
#define MC_GROUP_SHARDS   2

typedef struct mc_group_info {
    uint32_t n_members;        /* contexts (normally one per device)                              */
    uint32_t mode;             /* MC_GROUP_REPLICAS or MC_GROUP_SHARDS, as chosen at load          */
    uint32_t shard_kind;       /* 0 = none, 1 = minimizer line ranges, 2 = bucket ranges           */
    uint32_t peer_access;      /* 1: every pair of distinct devices has direct peer access         */
    uint64_t n_keys;           /* k-mers of the table                                              */
    uint64_t device_bytes_max; /* largest per-device share of the database in HBM                  */
    uint64_t bytes_needed_one; /* estimate used for the choice: HBM a full replica needs           */
    uint64_t bytes_free_min;   /* smallest free HBM among the devices when the choice was made     */
    uint32_t n_shards;         /* S: parts the table is cut into (1 = replicas)                    */
    uint32_t n_groups;         /* G: groups of S members that each hold the whole table; batches are dealt round-robin
                                  over the groups, rows are exchanged inside a group; S * G <= n_members              */
    uint32_t n_cycles;         /* C: 1 = the table is resident; > 1: it is larger than all devices together and cut into
                                  C * n_members parts, of which the members hold n_members at a time (mc_group_set_cycle)  */
    uint32_t cycle;            /* the cycle whose parts are loaded now                                                  */
} mc_group_info;

/* replaces: CuClarkDB::CuClarkDB device discovery + peer access (CuClarkDB.cu:118-215).
 * devices == NULL: the first n_devices visible devices, n_devices == 0: all of them (:146-150).
 * A device may be listed more than once (several contexts on one card: how the sharded path is
 * rehearsed and tested on a one-GPU box); with devices == NULL the environment variable
 * MC_GROUP_DEVICES=0,0,1 supplies such a list.  MC_ENODEVICE when fewer are visible than asked for (:140-144). */
int mc_group_open(mc_group **out, const int *devices, uint32_t n_devices, uint32_t k, uint64_t htsize,
                  uint32_t num_targets, uint32_t maxhits);
int mc_group_close(mc_group *g);

/* replaces: CuClarkDB::read for all devices (CuClarkDB.cu:463-770), incl. the memory budget that
 * decides how the table is cut (:516-559).  The files are streamed once per build pass and fed to all
 * devices.  mode: MC_GROUP_AUTO / _REPLICAS / _SHARDS (environment MC_GROUP_MODE=replicas|shards
 * overrides AUTO; MC_GROUP_HBM_BYTES caps the per-device memory the choice assumes; MC_GROUP_PARTS=S fixes the
 * part count of a sharded table).  Capacity: a part addresses 2^32 - 16 lines of its own (csrc/mc_minimizer.hpp
 * part_of), so N cards hold what their HBM holds: about 12e9 k-mers per 288 GB card at the densest fill.
 * AUTO that finds its estimate too kind (MC_ENOMEM while loading) cuts the table into more parts and tries again;
 * when the minimizer lines fit in no cut, the bucket-line table is the last resort (whole on every member when it fits
 * one, else the reference's bucket ranges, CuClarkDB.cu:552-559; mc_db_info.index_fallback = 1, said on stderr).
 * A table LARGER THAN ALL DEVICES TOGETHER is cut into C * N parts (by minimizer, as above) of which the N members hold N
 * at a time: cycle 0 is loaded here, mc_group_info.n_cycles says C, and the caller classifies every batch once per
 * cycle (mc_group_set_cycle, MC_F_FOLLOWUP) -- the reference's swapDbParts loop (CuClarkDB.cu:775-815,
 * src/CuCLARK_hh.hh:1765-1772); every change of cycle reads the database files again.  MC_GROUP_CYCLES=C forces C cycles
 * (tests).  More than 15 members, a table without a minimizer index, or parts that fit in no number of cycles up to 64:
 * MC_ENOMEM, "use more devices". */
int mc_group_load_db(mc_group *g, const char *base, int key_bytes, uint32_t sampling, int mode);
int mc_group_get_info(mc_group *g, mc_group_info *out);
/* replaces: CuClarkDB::swapDbParts (CuClarkDB.cu:775-815).  Loads the parts of cycle `cycle` (0 <= cycle < n_cycles) into the
 * members; nothing happens when they are loaded already.  All submitted batches must have been waited for.  The protocol of a
 * file (src/CuCLARK_hh.hh:1737-1772): cycle 0, every batch with MC_F_ROWS; cycle c > 0, every batch again with
 * MC_F_ROWS | MC_F_FOLLOWUP and the sparse rows it got in cycle c - 1 in its sparse-row buffer (they are merged with what the
 * parts of cycle c find and written back); MC_F_FINAL in the last cycle gives the final rows. */
int mc_group_set_cycle(mc_group *g, uint32_t cycle);
/* the member contexts, for mc_get_db_info / mc_get_stats */
int mc_group_member(mc_group *g, uint32_t i, mc_ctx **out);

/* as mc_alloc_batches / mc_batch_buffers / mc_submit / mc_wait / mc_free_batches, for the group:
 * replaces malloc, readyBatch + queryBatch, waitForBatch, freeBatchMemory (CuClarkDB.cu:321-421,
 * :820-987, :441-446, :284-316) */
int mc_group_alloc_batches(mc_group *g, uint32_t n_batches, uint64_t max_reads, uint64_t max_containers,
                           int want_rows);
int mc_group_batch_buffers(mc_group *g, uint32_t batch, uint32_t **reads_ptr, uint16_t **containers,
                           uint16_t **final_rows, uint16_t **sparse_rows);
int mc_group_submit(mc_group *g, uint32_t batch, uint64_t n_reads, uint64_t n_containers, uint32_t flags);
int mc_group_wait(mc_group *g, uint32_t batch);
int mc_group_sync(mc_group *g);
int mc_group_free_batches(mc_group *g);      /* the text buffers below as well */

/* as mc_text_alloc / _buffers / _submit / _wait (mc_api.h: FASTQ text in, final rows out; an addition to the reference's
 * interface), for a group whose members each hold the whole table: buffer b lives on member b % N.  MC_ESTATE for a table
 * that is cut into parts -- the caller then packs on the host as before. */
int mc_group_text_alloc(mc_group *g, uint32_t n_buffers, uint64_t max_text_bytes, uint64_t max_reads, uint64_t max_containers);
int mc_group_text_buffers(mc_group *g, uint32_t buffer, uint32_t **header_offsets, uint32_t **sequence_lengths, uint16_t **final_rows);
int mc_group_text_submit(mc_group *g, uint32_t buffer, const uint8_t *text, uint64_t n_bytes);
int mc_group_text_wait(mc_group *g, uint32_t buffer, uint64_t *n_reads, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* MC_GROUP_H */
