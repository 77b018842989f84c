/*
 * artp_c.h -- C ABI of libartp.so: the MI355X (gfx950) implementation of art_planner's
 * sampling + validity + edge hot path.
 *
 * This is the drop-in boundary.  The reference has no C/FFI seam for this path; its
 * seams are C++ virtuals.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference tree); art_planner_amd/host/ holds the C++ classes with the
 * reference's names/signatures that forward to these calls.
 *
 * Conventions
 *   - opaque context per device; all entry points return 0 (ARTP_OK) or a negative artp_status and
 *     never throw; calls on one context are serialised internally (one recursive lock held for the whole
 *     call, staging buffers included), contexts are independent.
 *   - plain pointers and sizes only.  Pointers are HOST memory unless the function name ends in
 *     `_dev`, in which case every buffer argument is DEVICE memory on the context's GPU and the call
 *     is asynchronous on the context's stream (artp_set_stream / artp_synchronize).
 *   - states are OMPL SE3 states flattened as 7 doubles: x y z qx qy qz qw.
 *   - dPose is the reference's HeightMapBoxChecker::dPose: 16 floats = origin[4], rotation[12]
 *     (row-major 3x4), art_planner/include/art_planner/validity_checker/height_map_box_checker.h:20-25.
 *   - layers are grid_map::Matrix storage: float32, column-major, rows = size.x, cols = size.y.
 */
#ifndef ARTP_C_H
#define ARTP_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct artp_ctx artp_ctx;

typedef enum {
  ARTP_OK = 0,
  ARTP_ERR_INVALID_ARG = -1,
  ARTP_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture: the product has NO CPU fallback */
  ARTP_ERR_HIP = -3,         /* a HIP runtime call failed; artp_last_error() has the text */
  ARTP_ERR_NO_MAP = -4,      /* a required layer was not uploaded (reference: hasMap() == false) */
  ARTP_ERR_CAPACITY = -5,    /* box too large for the LDS window tile of this context */
  ARTP_ERR_NO_WEIGHTS = -6,
  ARTP_ERR_TIMEOUT = -7,     /* artp_group_synchronize: a member's stream did not finish in time */
  ARTP_ERR_COMM = -8,        /* RCCL missing, or a communicator call failed (artp_group_last_error) */
  ARTP_ERR_COST_FUNC = -9    /* the caller's motion-cost function (artp_cost_set_external_query) reported failure */
} artp_status;

/* Numeric fields of art_planner::Params the hot path reads
 * (art_planner/include/art_planner/params.h:14-123).  Angles in radians. */
typedef struct {
  double torso_length, torso_width, torso_height;   /* params.h:93-95  */
  double torso_off_x, torso_off_y, torso_off_z;     /* params.h:97-101 */
  double feet_off_x, feet_off_y, feet_off_z;        /* params.h:107-111 */
  double reach_x, reach_y, reach_z;                 /* params.h:113-117 */
  int unknown_space_untraversable;                  /* params.h:26 */
  double max_pitch_pert, max_roll_pert;             /* params.h:80-81 */
  int sample_from_distribution;                     /* params.h:81: 1 = samplePositionInMapFromDist (default),
                                                       0 = samplePositionInMap (uniform in the map) */
} artp_params;

void artp_params_defaults(artp_params* p);          /* params.h defaults */
void artp_params_yaml(artp_params* p);              /* art_planner_ros/config/params.yaml:55-71 */

const char* artp_status_string(int status);
const char* artp_last_error(const artp_ctx* ctx);
/* "gfx950" etc. of the context's device. */
const char* artp_device_arch(const artp_ctx* ctx);

/* Planner / StateValidityChecker construction (art_planner/src/planner.cpp:75-131,
 * validity_checker.cpp:9-15).  Fails with ARTP_ERR_NO_DEVICE when there is no GPU. */
int artp_create(int device, const artp_params* params, artp_ctx** out);
void artp_destroy(artp_ctx* ctx);
/* Run the context's work on a caller-provided hipStream_t (NULL = HIP's legacy default stream).
 * A fresh context uses a private non-blocking stream; artp_use_own_stream switches back to it. */
int artp_set_stream(artp_ctx* ctx, void* hip_stream);
int artp_use_own_stream(artp_ctx* ctx);
int artp_synchronize(artp_ctx* ctx); /* every lane's stream */

/* Lanes: a context owns up to 4 lanes, each with its own stream (artp_set_stream applies to the current lane) and
 * its own scratch buffers, queues and counters; the map, its tables and the sampler are shared.  Calls issued on
 * different lanes overlap on the GPU: a planner thread that validates a batch as two halves on two lanes hides the
 * pipeline's short serial kernels and pairs kernels with different bottlenecks (bench.py: 1.77 -> 1.66 ms per 2^22
 * states).  Every call works on the CURRENT lane (0 after artp_create); artp_set_lane switches it (first use of a
 * lane creates its stream).  Work on different lanes is unordered: the caller orders map / layer updates against the
 * lanes that still read the old map (artp_synchronize, or stream events).  The current lane is state of the
 * context, not of the calling thread: lanes are for ONE host thread that keeps several batches in flight (threads
 * that share a context would switch each other's lane; give each thread its own context).  No equivalent in the
 * reference (it validates one state at a time on the calling thread). */
int artp_set_lane(artp_ctx* ctx, int lane);
int artp_get_lane(artp_ctx* ctx);

/* ---- map upload: HeightMapBoxChecker::setHeightField
 *      (art_planner/src/validity_checker/height_map_box_checker.cpp:38-54) ----------------------
 * slot 0 = body checker's layer (params.planner.elevation_layer, validity_checker_body.cpp:52-55),
 * slot 1 = feet checker's layer ("elevation_masked", validity_checker_feet.cpp:80-83).
 * Both slots share the grid geometry (also used for grid_map isInside). */
enum { ARTP_SLOT_BODY = 0, ARTP_SLOT_FEET = 1 };
int artp_upload_layer(artp_ctx* ctx, int slot, const float* layer_colmajor, int rows, int cols,
                      double len_x, double len_y, double pos_x, double pos_y);
/* Config-5 incremental update: overwrite the rectangle [row0,row0+nrows) x [col0,col0+ncols) of a
 * previously uploaded layer; `patch` is column-major nrows x ncols. */
int artp_update_layer_rect(artp_ctx* ctx, int slot, const float* patch, int row0, int col0,
                           int nrows, int ncols);
/* Several rectangles of one slot in one call (what computeChange, change.cpp:9-51, yields per map update):
 * patches[k] is column-major nrows x ncols, rects = n_rects x {row0, col0, nrows, ncols}, applied IN ORDER (where
 * rectangles overlap the later one wins).  One staged copy, the range tables rebuilt once, the partner table (kept as
 * counts) re-evaluated for the pairs with an end in a changed rectangle only.  Both forms are ASYNCHRONOUS on the context's stream: the
 * patches are copied to a pinned staging buffer before the call returns (the caller's buffers are free again), the
 * device work is ordered in front of whatever the stream runs next. */
int artp_update_layer_rects(artp_ctx* ctx, int slot, int n_rects, const float* const* patches, const int* rects);
/* Map version of the context: a counter that every call which changes what isValid() / the sampler would answer
 * increments -- artp_upload_layer, artp_update_layer_rect, artp_upload_sampler_layers, artp_preprocessed_install,
 * artp_preprocessed_reweight_dev (when it installs the new distribution).  The reference's guarantee that
 * StateValidityChecker::updateHeightField (validity_checker.cpp:26-29) takes effect for the very next isValid() is
 * what a host-side label cache has to honour: it stamps cached labels with the version they were computed on
 * (artp_sample_and_validate reports it) and compares with artp_map_version() before serving one.  Lock-free read. */
uint64_t artp_map_version(const artp_ctx* ctx);

/* ---- HeightMapBoxChecker::checkCollision (height_map_box_checker.cpp:58-72) -------------------
 * hit[i] = (dCollide(box, field, 1, ...) != 0) for pose i.  exit_codes (optional, may be NULL)
 * receives which early-out of dCollideHeightfieldZone decided (ARTP_EXIT_*). */
enum {
  ARTP_EXIT_AABB_OFF = 0, ARTP_EXIT_ABOVE = 1, ARTP_EXIT_UNDER = 2, ARTP_EXIT_SPANS = 3,
  ARTP_EXIT_FLAT_PLANE = 4, ARTP_EXIT_VERTEX = 5, ARTP_EXIT_PLANE = 6, ARTP_EXIT_VERTEX2 = 7,
  ARTP_EXIT_NONE = 8
};
int artp_check_boxes(artp_ctx* ctx, int slot, const float box_lengths[3], const float* dposes,
                     size_t n, uint8_t* hit, uint8_t* exit_codes);
int artp_check_boxes_dev(artp_ctx* ctx, int slot, const float box_lengths[3], const float* dposes,
                         size_t n, uint8_t* hit, uint8_t* exit_codes);

/* ---- ob::StateValidityChecker::isValid, batched
 *      (art_planner/src/validity_checker/validity_checker.cpp:39-45) ----------------------------
 * valid[i] = body_ok && feet_ok for state i.  detail (optional): n x 6 int8 =
 * {body exit code | -1 outside map, 4 x foot exit code | -1 outside | -2 not evaluated, 0}. */
/* Up to 16 states without `detail` take the latency path: one launch, one workgroup per state with the five
 * boxes side by side, states / labels through mapped pinned host memory (no copies). */
int artp_validate_states(artp_ctx* ctx, const double* se3, size_t n, uint8_t* valid, int8_t* detail);
/* Persistent latency service for the per-state isValid() of the host mirror (validity_checker.cpp:39-45: one OMPL call =
 * one state).  enabled != 0: artp_validate_states calls with one or two states and no `detail` are answered by ONE RESIDENT
 * workgroup that polls a mailbox in mapped host memory -- no kernel launch per call.  The workgroup is started on demand on a
 * stream of its own, restarted after every map write (artp_map_version) and leaves by itself after 200 us without a
 * request (2 s at most in all): it cannot outlive a host that stopped calling, and a burst's end frees the device before
 * a hipFree / hipDeviceSynchronize (which wait for every stream) can trip over it.  Same labels as the launch-per-call path
 * (the same device function).  Off by default: it holds five wavefronts and ~63 KB of LDS of one CU while resident.
 * The same switch covers ob::MotionValidator::checkMotion one edge at a time (prm_motion_cost.cpp:652,
 * lazy_prm_star_min_update.cpp:725, OMPL's PathSimplifier): artp_check_motions / artp_check_motions_last_valid /
 * artp_check_edges_interp calls with one or two edges (host pointers; at most 8 x 256 interpolation states) are answered by a
 * resident POOL of 256 workgroups (one per CU) with the same start / restart / idle rules.  Every workgroup polls a request block in
 * device memory that the host writes through the PCIe BAR, runs its share of the edge's states and reports to a slot of its
 * own in mapped host memory; the host reduces the slots.  (A device that does not expose its memory to the host -- no large
 * BAR -- has no pool: these calls keep their one launch per call.)
 * Same verdicts, lastValid pairs and interpolation counts as the other paths, bit for bit.
 * artp_persistent_latency_stats: out[0] = launches of the resident kernels so far, out[1] = requests they answered. */
int artp_set_persistent_latency(artp_ctx* ctx, int enabled);
int artp_persistent_latency_stats(artp_ctx* ctx, uint64_t out[2]);
int artp_validate_states_dev(artp_ctx* ctx, const double* se3, size_t n, uint8_t* valid,
                             int8_t* detail);

/* ---- ob::StateSampler::sampleUniform, batched
 *      (art_planner/src/sampler.cpp:56-131; SE3FromSE2Sampler) ----------------------------------
 * The seven layers the sampler reads (sampler.cpp:61-63,99-103; map.h:64-116), same geometry as the
 * validity layers.  cum_prob_rowwise = column 0 of "cum_prob_rowwise_hack" (rows floats). */
int artp_upload_sampler_layers(artp_ctx* ctx, const float* cum_prob, const float* cum_prob_rowwise,
                               const float* elevation, const float* normal_x, const float* normal_y,
                               const float* normal_z, const float* plane_fit_std_dev, int rows,
                               int cols, double len_x, double len_y, double pos_x, double pos_y);
/* State k of the batch is a pure function of (seed, first_index + k): six counter-based uniforms in
 * the reference's draw order replace ompl::RNG's mt19937 stream (SURVEY.md 8c "same seed"). */
int artp_sample_states(artp_ctx* ctx, uint64_t seed, uint64_t first_index, size_t n, double* se3_out);
int artp_sample_states_dev(artp_ctx* ctx, uint64_t seed, uint64_t first_index, size_t n,
                           double* se3_out);
/* Fused rejection-sampling step of the planners' hot loop
 * (prm_motion_cost.cpp:174-186, lazy_prm_star_min_update.cpp:552-554): sample n states, validate
 * them, write states + labels.  n_valid (host pointer, optional) receives the number of valid ones
 * (forces a stream sync when non-NULL). */
int artp_sample_and_validate_dev(artp_ctx* ctx, uint64_t seed, uint64_t first_index, size_t n,
                                 double* se3_out, uint8_t* valid_out, size_t* n_valid);
/* Host-buffer form: states and labels come back together, so a sampler that hands the states out one at a
 * time (ob::StateSampler::sampleUniform) already holds the label the following isValid() will ask for.
 * map_version (optional): the artp_map_version() the states were sampled and validated on (the context's lock is
 * held for the whole call, so it is the version of every label of the batch). */
int artp_sample_and_validate(artp_ctx* ctx, uint64_t seed, uint64_t first_index, size_t n, double* se3_out,
                             uint8_t* valid_out, uint64_t* map_version);

/* ---- ob::MotionValidator::checkMotion (OMPL DiscreteMotionValidator; call sites
 *      prm_motion_cost.cpp:652, lazy_prm_star_min_update.cpp:725), batched -----------------------
 * valid[i] = isValid(s2_i) && all interior states of the discretised segment are valid.
 * R^3 bounds for validSegmentCount follow planner.cpp:146-156; z bounds come from
 * artp_set_z_bounds (min/max finite elevation -/+ reach.z/2). */
int artp_set_z_bounds(artp_ctx* ctx, double z_low, double z_high);
/* checkMotion's R^3 maxExtent (the diagonal of the state space's x / y / z bounds; the segment length is 1 % of it)
 * fixed to max_extent instead of following the installed map and the z bounds; 0 = follow them (default).  For hosts that
 * want OMPL's own behaviour: Planner::setMap calls space_->setBounds (planner.cpp:146-163) but nothing re-runs
 * StateSpace::setup(), so longestValidSegment_ stays at the FIRST planned map's extents. */
int artp_set_r3_extent(artp_ctx* ctx, double max_extent);
int artp_check_motions(artp_ctx* ctx, const double* s1, const double* s2, size_t n, uint8_t* valid);
/* Latency form.  The reference calls checkMotion ONE edge at a time on the solution path (prm_motion_cost.cpp:652,
 * lazy_prm_star_min_update.cpp:725, OMPL's PathSimplifier via planner.cpp:272): the HOST-buffer entry points
 * (artp_check_motions, artp_check_motions_last_valid, artp_check_edges_interp) take n <= 64 edges in ONE kernel launch
 * -- edges and verdicts through mapped host memory, no copies, no host round trip for the segment counts.  Same
 * verdicts, lastValid pairs and error codes as the batch pipeline (tests/test_gpu_parity.py).  enabled = 0 sends small
 * calls through the batch pipeline like large ones (default 1). */
int artp_set_few_edges(artp_ctx* ctx, int enabled);
/* How large batches of artp_check_motions run (first overload): two_pass != 0 (default) = s2 and every coarse_stride-th
 * interior state of every edge first, the rest only for the edges still alive (an edge is valid iff all its states are, so
 * the verdicts do not depend on it); two_pass == 0 = one pass over all states.  coarse_stride >= 2, 0 = the default (8). */
int artp_set_edge_passes(artp_ctx* ctx, int two_pass, int coarse_stride);
int artp_check_motions_dev(artp_ctx* ctx, const double* s1, const double* s2, size_t n,
                           uint8_t* valid);
/* ob::MotionValidator::checkMotion(s1, s2, std::pair<State*, double>& lastValid) (pure virtual in OMPL 1.4.2;
 * DiscreteMotionValidator tests the interior states j = 1 .. nd-1 in order, then s2): for a failing edge
 * last_valid_t[i] = lastValid.second = (j - 1) / nd of the first failing j ((nd - 1) / nd when only s2 fails) and
 * last_valid_se3[i] (n x 7, may be NULL) = interpolate(s1, s2, last_valid_t[i]) = *lastValid.first; a passing
 * edge reports t = 1 and s2 (OMPL leaves lastValid untouched). */
int artp_check_motions_last_valid(artp_ctx* ctx, const double* s1, const double* s2, size_t n, uint8_t* valid,
                                  double* last_valid_t, double* last_valid_se3);
int artp_check_motions_last_valid_dev(artp_ctx* ctx, const double* s1, const double* s2, size_t n, uint8_t* valid,
                                      double* last_valid_t, double* last_valid_se3);
/* PRM-build edge validation of PRMMotionCost::addValidMilestone (prm_motion_cost.cpp:340-377):
 * n_interp = floor(lateral distance / 0.5) interior states at t = step * (1/(n_interp+1)), each
 * must be valid.  n_interp_out optional (uint32 per edge). */
int artp_check_edges_interp(artp_ctx* ctx, const double* s1, const double* s2, size_t n,
                            uint8_t* valid, uint32_t* n_interp_out);
int artp_check_edges_interp_dev(artp_ctx* ctx, const double* s1, const double* s2, size_t n,
                                uint8_t* valid, uint32_t* n_interp_out);

/* ---- batch plumbing around the hot path ---------------------------------------------------------
 * Stream compaction of the valid states (the planners keep only accepted samples,
 * prm_motion_cost.cpp:174-187): out_se3 receives the states with valid[i] != 0 in input order,
 * *n_out_dev (device uint64) their number.  out_se3 must hold n states. */
int artp_compact_valid_dev(artp_ctx* ctx, const double* se3, const uint8_t* valid, size_t n,
                           double* out_se3, uint64_t* n_out_dev);
/* Multi-GPU exchange helpers.  A state is a pure function of (seed, sample index), so ranks exchange the
 * 4-byte indices of their accepted states (RCCL all-gather) and materialise remote states locally:
 * out_idx receives the positions i < n with valid[i] != 0 in order, *n_out_dev their number;
 * artp_sample_states_at_dev writes the states of indices base_index + idx[j], j < min(*count_dev, cap). */
int artp_compact_valid_indices_dev(artp_ctx* ctx, const uint8_t* valid, size_t n, uint32_t* out_idx,
                                   uint64_t* n_out_dev);
int artp_sample_states_at_dev(artp_ctx* ctx, uint64_t seed, uint64_t base_index, const uint32_t* idx,
                              const uint64_t* count_dev, size_t cap, double* se3_out);
/* The most compact form of "which states of my batch were accepted": one bit per candidate (512 KiB for 2^22
 * candidates, against 9 MB of indices or 130 MB of states) -- what the ranks all-gather per batch.
 * bits_out: ceil(n / 64) 64-bit words, bit k of word w = valid[64 w + k] != 0.  artp_indices_from_bits_dev turns
 * the first n bits of a (received) bitmap back into the ascending index list artp_sample_states_at_dev takes. */
int artp_pack_valid_bits_dev(artp_ctx* ctx, const uint8_t* valid, size_t n, uint64_t* bits_out);
/* The receiving side in one call for ALL ranks: bits = n_ranks bitmaps of words_per_rank words each (the all-gather's
 * output), base_index[r] (host array) = first global sample index of rank r's batch.  For every rank the first
 * min(count, cap) accepted states among its first prefix_bits candidates are re-sampled into
 * se3_out[r * cap .. ) (n_ranks x cap x 7 doubles, device); counts_dev[r] (device) = accepted states among the
 * prefix.  Two launches whatever n_ranks is (<= 16). */
int artp_materialise_from_bits_dev(artp_ctx* ctx, uint64_t seed, const uint64_t* bits, int n_ranks,
                                   size_t words_per_rank, size_t prefix_bits, const uint64_t* base_index, size_t cap,
                                   double* se3_out, uint64_t* counts_dev);
int artp_indices_from_bits_dev(artp_ctx* ctx, const uint64_t* bits, size_t n, uint32_t* out_idx, uint64_t* n_out_dev);
/* The second exchange of SURVEY.md 8e: the edge results of a rank as fixed-size records {u32 i, u32 j,
 * f32 cost[3]} (20 bytes; i / j = the caller's vertex ids of the edge's endpoints, cost = the MotionCostFunc row
 * of artp_cost_query_dev or any 3 floats).  records_out (n x 5 u32) receives the records of the edges with
 * valid[e] != 0 in input order, *n_out_dev their number; the block is then all-gathered (RCCL). */
int artp_pack_edge_results_dev(artp_ctx* ctx, const uint8_t* valid, const uint32_t* edge_i, const uint32_t* edge_j,
                               const float* cost, size_t n, uint32_t* records_out, uint64_t* n_out_dev);
/* ---- multi-GPU: device groups --------------------------------------------------------------
 * The path shards over independent sample batches: the map is replicated, rank r of W owns the states
 * [ (step W + r) S, (step W + r + 1) S ) of the (seed, index) sample stream (artp_shard_first_index: gap-free for
 * any W, so labels do not depend on W), and the exchange steps are all-gathers of fixed-size blocks over xGMI.
 * A group owns, for every LOCAL device, one artp_ctx, one RCCL communicator (librccl is bound at run time with
 * dlopen: libartp.so itself does not depend on it), a side stream for the collectives and the double-buffered
 * exchange buffers.  This is what lets a C++ host -- the reference's PlannerRos keeps ONE planner object and calls
 * it from its planning thread, art_planner_ros/src/planner_ros.cpp:250-319 -- use every GPU of the node without an
 * interpreter in the process.  Two ways to build one:
 *   artp_group_create       one process, n devices (ncclCommInitAll); every device gets a host worker thread, so the
 *                           launches of a step are issued in parallel.
 *   artp_group_create_rank  one process per GPU (torchrun / mpirun style): rank 0 makes the id
 *                           (artp_group_unique_id), the launcher distributes its 128 bytes, every rank joins
 *                           (ncclCommInitRank).
 * transport: ARTP_GROUP_RCCL (default) or ARTP_GROUP_PEER_COPY -- single-process groups only: the all-gather as
 * hipMemcpyPeerAsync pushes between the members; it also accepts the SAME device several times, which is how the
 * W > 1 sharding / exchange logic is tested on a one-GPU box.
 * The map: upload / install it on every member's context (artp_group_ctx) -- maps are replicated.  A group call
 * returns the first failing member's status; artp_group_last_error has the text. */
typedef struct artp_group artp_group;
enum { ARTP_GROUP_RCCL = 0, ARTP_GROUP_PEER_COPY = 1 };
#define ARTP_GROUP_ID_BYTES 128
uint64_t artp_shard_first_index(uint64_t step, int rank, int world, uint64_t batch);
int artp_group_create(const int* devices, int n, const artp_params* params, int transport, artp_group** out);
int artp_group_unique_id(uint8_t id[ARTP_GROUP_ID_BYTES]);
int artp_group_create_rank(int device, int rank, int world, const uint8_t id[ARTP_GROUP_ID_BYTES],
                           const artp_params* params, artp_group** out);
void artp_group_destroy(artp_group* g);
const char* artp_group_last_error(const artp_group* g);
int artp_group_world_size(const artp_group* g);
int artp_group_local_count(const artp_group* g);
int artp_group_rank(const artp_group* g, int local);           /* global rank of local member `local` */
artp_ctx* artp_group_ctx(artp_group* g, int local);            /* owned by the group */
/* The number of ranks the transport itself sees: an all-reduce (sum) of ones. */
int artp_group_ranks_seen(artp_group* g, int* ranks_seen);
/* Sizes of the state exchange: batch = S candidates per rank and step; every rank's first
 * min(accepted, materialise_cap) accepted states among its first prefix candidates (0 = all S) are re-materialised
 * on every member after the all-gather (0 = bitmaps only).  (Re)allocates the buffers; drains the group first -- for at
 * most $ARTP_GROUP_CONFIGURE_TIMEOUT_MS milliseconds (default 60000; -1 = wait for ever; a value that is not an integer
 * >= -1 keeps the default), after which the call returns ARTP_ERR_TIMEOUT with nothing reallocated. */
int artp_group_configure(artp_group* g, uint64_t seed, size_t batch, size_t materialise_cap, size_t prefix);
/* One step of the sharded rejection-sampling loop (prm_motion_cost.cpp:174-186 / lazy_prm_star_min_update.cpp:552-554
 * across the node), ASYNCHRONOUS: for every local member -- artp_sample_and_validate_dev on its shard
 * [artp_shard_first_index(step, rank, W, S), + S), one validity bit per candidate (artp_pack_valid_bits_dev), the
 * all-gather of the W bitmaps on the side stream, artp_materialise_from_bits_dev behind it.  Buffers are
 * double-buffered by step parity: step k + 2 waits for step k's exchange on the device, the host never blocks. */
int artp_group_sample_and_validate_step(artp_group* g, uint64_t step);
/* What a step left on member `local` (device pointers, valid once the step's work is done: artp_group_synchronize):
 * se3 / valid = its own S candidates and labels (single-buffered: the next step overwrites them), bits = the W
 * gathered bitmaps (W x ceil(S / 64) words), states = W x materialise_cap x 7 doubles, counts = W accepted counts
 * among the prefix.  Any out pointer may be NULL. */
int artp_group_step_buffers(artp_group* g, int local, uint64_t step, const double** se3, const uint8_t** valid,
                            const uint64_t** bits, const double** states, const uint64_t** counts);
/* The second exchange: per local member the n edges it owns -- valid / edge_i / edge_j / cost as for
 * artp_pack_edge_results_dev (device pointers, on that member's GPU) -- packed into 20-byte records and all-gathered
 * in blocks of `cap` records (n <= cap on every rank; the same cap on every rank).  ASYNCHRONOUS.  Result on every
 * member (artp_group_edge_buffers): records = W x cap x 5 u32, counts = W record counts.  i / j stay in the owner's
 * numbering; add artp_shard_first_index(step, r, W, S) for block r on arrival. */
typedef struct artp_group_edges {
  const uint8_t* valid; const uint32_t* edge_i; const uint32_t* edge_j; const float* cost; size_t n;
} artp_group_edges;
int artp_group_exchange_edges(artp_group* g, const artp_group_edges* per_local, size_t cap);
int artp_group_edge_buffers(artp_group* g, int local, const uint32_t** records, const uint64_t** counts);
/* Waits until every stream of every local member is idle, at most timeout_ms (< 0: no limit).  ARTP_ERR_TIMEOUT
 * names the member and the stream that did not finish (a collective whose peer never arrived) and leaves the group
 * as it is: artp_group_abort tears the communicators down (ncclCommAbort) so that the process can report and exit. */
int artp_group_synchronize(artp_group* g, int timeout_ms);
int artp_group_abort(artp_group* g);

/* Measurement helper: the ALGORITHMIC window size of a batch = sum over states of
 * the heightfield vertices (nMaxX-nMinX+1)*(nMaxZ-nMinZ+1) of all five boxes, no credit for
 * early-outs or short-circuiting, 0 for a box whose centre is outside the map or whose AABB is off
 * the field.  Host result. */
int artp_algorithmic_vertices_dev(artp_ctx* ctx, const double* se3, size_t n, uint64_t* total_vertices);
/* Diagnostics of the last validity batch: out[0] = torso boxes queued for the window stage,
 * out[4] = foot boxes queued, out[1] = boxes that needed the exact plane grouping. */
int artp_debug_pipeline_counters(artp_ctx* ctx, uint64_t out[8]);
/* Diagnostics: the partner table of a layer (one byte per sample in ODE sample layout, bit 0 / bit 1 =
 * the ABC / DBC triangle of the cell has an epsilon-equal plane within *radius cells).
 * out may be NULL to query the radius only; *radius = 0 when the layer has no table. */
int artp_debug_partner_table(artp_ctx* ctx, int slot, uint8_t* out, size_t out_bytes, int* radius);

/* ---- "next" row N1: batched roadmap front end ------------------------------------------
 * Replaces, as one batch per stage, the sampling / connection / search loops of the reference planners:
 *   PRMMotionCostMaintainer::sampleGraph  art_planner/src/planners/prm_motion_cost.cpp:145-219
 *   PRMMotionCost::addValidMilestone      prm_motion_cost.cpp:325-390 (k nearest, 0.5 m interpolation rule)
 *   PRMMotionCost::constructSolution      prm_motion_cost.cpp:536-673 (A*, lazy checkMotion of the path)
 *   PathLengthObjective                   art_planner/src/objectives/path_length_objective.cpp:26-70
 * Vertex 0 = start, vertex 1 = goal, vertices 2.. = the first n_milestones accepted states of the
 * (seed, index) sample stream.  Needs both height layers, the sampler layers and artp_set_z_bounds. */
typedef struct artp_roadmap artp_roadmap;
struct artp_preprocessed;        /* "next" row N2 below */
struct artp_preprocess_params;
typedef struct artp_roadmap_params {
  uint64_t seed, first_index;  /* sample stream */
  uint32_t n_milestones;       /* Params::planner.prm_motion_cost.max_n_vertices (params.h:51) */
  uint32_t k_neighbors;        /* 0 = OMPL KStarStrategy: ceil(e (1 + 1/6) ln n_vertices) */
  int32_t objective;           /* 0 = PathLengthObjective::motionCostHeuristic (Euclidean / max_lon_vel),
                                  1 = directional time cost (use_directional_cost, params.h:70),
                                  2 = learned motion cost: PRMMotionCostMaintainer::updateEdges
                                      (prm_motion_cost.cpp:27-73) over artp_cost_query; needs
                                      artp_cost_load_weights + artp_cost_update_map */
  uint32_t max_replans;        /* bound on lazy edge removals in artp_roadmap_solve */
  double max_lon_vel, max_lat_vel, max_ang_vel; /* params.h:71-73 */
  float w_energy, w_time, w_risk; /* MotionCostObjective::getCost weights (params.h:58-62) */
  float risk_threshold;           /* MotionCostObjective::isFeasible (params.h:55) */
  /* PRMMotionCostMaintainer::sampleGraph's budgets and its in-build re-weighting (prm_motion_cost.cpp:171-193):
   * the graph grows until max_n_vertices (= n_milestones) OR max_n_edges candidate edges OR max_sample_time of
   * sampling; every recompute_density_after_n_samples accepted vertices Map::reApplyPreprocessing() recomputes the
   * sampling distribution from the inverse density of the vertices so far.  0 switches a budget / the re-weighting
   * off.  The re-weighting needs the preprocessing result of the installed map (density_map, artp_preprocess_map_ex)
   * and its parameters; with density_map == NULL the distribution stays fixed. */
  uint32_t max_n_edges;                        /* params.h:52 (default 50000) */
  uint32_t recompute_density_after_n_samples;  /* params.h:53 (default 1000) */
  double max_sample_time;                      /* seconds, params.h:50 (default 2.0) */
  struct artp_preprocessed* density_map;
  const struct artp_preprocess_params* density_params;
  /* How the graph is put together from the accepted-state stream (same stream, same predicates in every mode):
   *   0 = batched (default): all milestones first, k nearest over the FINAL vertex set, symmetrised pairs, one
   *       interpolation-rule verdict and one chain cost per pair.  Fastest; not the reference's graph.
   *   1 = PRMMotionCost::addValidMilestone order (prm_motion_cost.cpp:325-390), the reference's OWN graph: milestones
   *       are inserted one at a time, each is connected to the k = ceil(e (1 + 1/6) ln n) nearest vertices present
   *       at ITS insertion (n = vertices then, itself included; it becomes a neighbour target last), the VALID
   *       PREFIX of every 0.5 m chain stays in the graph as vertices that are neighbour targets for later milestones
   *       (:353-371), start and goal join last (baseSolve, :447-470).  n_milestones is the reference's
   *       max_n_vertices and counts the chain vertices too, max_n_edges counts the sub-edges (:171-172).  The
   *       insertion loop is sequential by construction (host); the interior states of a milestone's k chains are
   *       interpolated and validated as one small device batch.  Vertex ids: 0 start, 1 goal, then insertion order.
   *   2 = LazyPRMStarMinUpdate::addValidMilestone order (lazy_prm_star_min_update.cpp:424-446), BASELINE config 1's
   *       planner: start, goal, then the milestones; vertex i gets DIRECT edges of unknown validity to the
   *       k = ceil(e (1 + 1/6) ln (i + 1)) nearest of its predecessors, validity is established lazily by
   *       artp_roadmap_solve.  Predecessor-only search with a per-vertex k is one batch on the device.
   * In modes 1 and 2 an edge's cost is evaluated ONCE in the direction the reference adds it to its undirected graph
   * (new vertex -> neighbour, along the chain from the milestone: opt_->motionCost(m, n), updateEdges' source ->
   * target) -- visible with the directional and the learned objective; mode 0 evaluates smaller id -> larger id. */
  int32_t construction;
  /* Params::planner.prm_motion_cost.max_query_edge_length (params.h:54, default 0.5): MotionCostObjective::motionCost
   * (motion_cost_objective.cpp:41-42) splits a motion into n_interp = (unsigned)(lateral distance / this) + 1 cost
   * queries -- what prices a path SEGMENT with the learned objective: the shortcut candidates of
   * artp_roadmap_simplify_path and the path costs it compares.  (The GRAPH's sub-edges are priced one query each by
   * PRMMotionCostMaintainer::updateEdges, and the 0.5 m of addValidMilestone's validity chain is a constant of its
   * own, prm_motion_cost.cpp:343: neither depends on this.)  0 = 0.5. */
  double max_query_edge_length;
} artp_roadmap_params;
void artp_roadmap_params_defaults(artp_roadmap_params* p);
/* Samples, connects and validates.  ARTP_ERR_INVALID_ARG (artp_last_error says which) when start or goal
 * is not a valid state (OMPL: INVALID_START / INVALID_GOAL, prm_motion_cost.cpp:452-476). */
int artp_roadmap_build(artp_ctx* ctx, const artp_roadmap_params* params, const double* start_se3,
                       const double* goal_se3, artp_roadmap** out);
/* out[0] vertices, [1] candidate edges, [2] edges passing the interpolation rule, [3] edges removed by
 * the lazy path check so far, [4] k, [5] samples drawn, [6] density re-weightings during the build,
 * [7] bit 0 = the sampling-time budget ended the build, bit 1 = the edge budget cut the vertex set back, bit 2 = the
 * graph is STILL over max_n_edges (the bounded prefix search gave up: treat the budget as not honoured). */
int artp_roadmap_stats(const artp_roadmap* rm, uint64_t out[8]);
/* Any pointer may be NULL.  verts: n_vertices x 7; knn / knn_dist: n_vertices x k (0xffffffff = none);
 * edges_uv: n_edges x 2 (u < v, sorted); edge_*: n_edges. */
int artp_roadmap_export(const artp_roadmap* rm, double* verts, uint32_t* knn, double* knn_dist,
                        uint32_t* edges_uv, uint8_t* edge_valid, uint32_t* edge_interp, double* edge_cost,
                        uint8_t* edge_removed);
/* Cheapest start -> goal path whose edges also pass the discrete motion validator (artp_check_motions).
 * *n_path = 0 when start and goal are not connected; ARTP_ERR_CAPACITY (with *n_path = needed) when
 * path_se3 (cap_states x 7, may be NULL) is too small. */
int artp_roadmap_solve(artp_roadmap* rm, double* path_se3, size_t cap_states, size_t* n_path, double* cost,
                       int* n_replans);
/* LazyPRMStarMinUpdate::baseSolve (lazy_prm_star_min_update.cpp:552-615): the roadmap keeps growing while the
 * planner has time -- solve, then grow by grow_step milestones (artp_roadmap_grow) and solve again until
 * plan_time seconds have passed; the best solution found is returned (the reference keeps bestSolution / bestCost_
 * the same way).  stats (may be NULL): [0] growth rounds, [1] final vertex count, [2] rounds that improved the cost.
 * *n_path = 0 when no round connected start and goal (PlannerStatus::TIMEOUT). */
int artp_roadmap_solve_until(artp_roadmap* rm, double plan_time, uint32_t grow_step, double* path_se3,
                             size_t cap_states, size_t* n_path, double* cost, uint64_t stats[3]);
/* After the map changed (artp_upload_layer / artp_update_layer_rect / artp_preprocessed_install): re-validate
 * every vertex and re-evaluate every edge against the current layers, forget earlier lazy removals -- the
 * batched form of LazyPRMStarMinUpdate's roadmap maintenance (lazy_prm_star_min_update.cpp:18-217).
 * out (may be NULL): [0] vertices now invalid, [1] / [2] edges passing the rule before / after,
 * [3] bit 0 = start still valid, bit 1 = goal still valid. */
int artp_roadmap_revalidate(artp_roadmap* rm, uint64_t out[4]);
/* New start / goal on the kept roadmap: vertices 0 and 1 are replaced and connected to their k nearest
 * vertices (every OMPL query adds its start and goal as milestones, prm_motion_cost.cpp:452-476). */
int artp_roadmap_set_query(artp_roadmap* rm, const double* start_se3, const double* goal_se3);
/* Solution simplification (Params::planner.simplify_solution; the reference calls OMPL's randomised
 * PathSimplifier through ss_->simplifySolution(), planner.cpp:266-280): every pair of path states is tried as a shortcut in one batch
 * (interpolation rule + discrete motion validator + the objective's cost) and the cheapest chain of valid
 * shortcuts is returned -- deterministic, never worse than the input.  out_se3 holds up to n states. */
int artp_roadmap_simplify_path(artp_roadmap* rm, const double* path_se3, size_t n, double* out_se3,
                               size_t* n_out, double* cost);
/* Growing the kept roadmap (PRMMotionCostMaintainer::sampleGraph keeps adding milestones between queries,
 * prm_motion_cost.cpp:145-219; LazyPRM* grows while it plans): the milestones still valid on the CURRENT map
 * stay, n_more new ones are drawn where the sample stream left off, connections (k follows the vertex count)
 * and edge verdicts are recomputed for the whole set.  out (may be NULL) = {milestones kept, milestones the
 * current map invalidated}.  Start and goal must still be valid (else ARTP_ERR_INVALID_ARG: set a new query). */
int artp_roadmap_grow(artp_roadmap* rm, uint64_t n_more, uint64_t out[2]);
/* The preprocessing result re-weightings are computed on (artp_roadmap_params::density_map) belongs to the map it
 * was made from: after a map update hand the roadmap the new one (or NULL) before growing it. */
int artp_roadmap_set_density_map(artp_roadmap* rm, struct artp_preprocessed* pp,
                                 const struct artp_preprocess_params* params);
void artp_roadmap_destroy(artp_roadmap* rm);

/* ---- "next" row N2: the per-map preprocessing chain on the device -----------------------
 * Replaces processors::Basic (art_planner/src/map/processors/basic.cpp:42-143: traversability threshold,
 * safety morphology, elevation_masked, sample filter), estimateNormals (art_planner/src/utils.cpp:213-326)
 * and computeCumulativeProbabilityDistribution (processors/probability_distribution.cpp:20-46), which the
 * reference runs on the CPU (OpenCV) on every map update.  Layers are grid_map matrices (float32,
 * column-major rows x cols).  Inpainting (inpaintMatrix, utils.cpp:13-64) stays with the caller: the
 * input layers must be hole-free. */
typedef struct artp_preprocessed artp_preprocessed;
typedef struct artp_preprocess_params {
  float traversability_thres;                    /* Params::planner.traversability_thres (params.h:24) */
  double foothold_margin;                        /* Params::planner.safety.* (params.h:27-34) */
  double foothold_margin_max_hole_size;
  double foothold_margin_max_drop;
  double foothold_margin_max_drop_search_radius;
  double foothold_margin_min_step;
  double foothold_size;
  int use_inverse_vertex_density;                /* Params::sampler.* (params.h:82-84, planner.cpp:43-55) */
  int use_max_prob_unknown_samples;
  double max_prob_unknown_samples;
} artp_preprocess_params;
typedef struct artp_preprocess_inputs { /* column-major rows x cols float layers on the host */
  const float* elevation;       /* required */
  const float* traversability;  /* NULL = all 1 (Basic::checkTraversability, basic.cpp:13-22) */
  const float* observed;        /* NULL = all 1; Basic::addKnownCells (basic.cpp:26-38): cells valid before inpainting */
  const double* vertex_se3;     /* roadmap vertices (n_vertices x 7) for computeInverseSampleDensity; may be NULL */
  size_t n_vertices;
  int rows, cols;
  double len_x, len_y, pos_x, pos_y;
} artp_preprocess_inputs;
void artp_preprocess_params_defaults(artp_preprocess_params* p); /* params.h defaults */
void artp_preprocess_params_yaml(artp_preprocess_params* p);     /* art_planner_ros/config/params.yaml */
/* elevation: required; traversability: NULL = all 1 (Basic::checkTraversability, basic.cpp:13-22).
 * The robot numbers (normal estimation radius, reach) come from the context's artp_params. */
int artp_preprocess_map(artp_ctx* ctx, const float* elevation, const float* traversability, int rows, int cols,
                        double len_x, double len_y, double pos_x, double pos_y,
                        const artp_preprocess_params* params, artp_preprocessed** out);
/* The whole new-map chain of Planner::setUpMapProcessors (planner.cpp:39-58): Basic, then -- when the params ask
 * for it -- computeInverseSampleDensity (sample_density.cpp:12-43: vertices per cell, cv::GaussianBlur of radius
 * (torso.length + torso.width) / 4, probability = max - blurred), applyBaseSampleDistribution,
 * applyMaxUnknownProbability (probability_distribution.cpp:50-90) and the CDF. */
int artp_preprocess_map_ex(artp_ctx* ctx, const artp_preprocess_inputs* in, const artp_preprocess_params* params,
                           artp_preprocessed** out);
/* The old-map chain: computeChange (processors/change.cpp:9-51) between two results of the same size (their
 * origins may differ by whole cells).  updated_out (rows x cols, may be NULL) receives the "updated" layer,
 * rect = {row0, col0, nrows, ncols} bounds the updated cells of the new map (nrows = 0: none). */
int artp_preprocessed_change(artp_ctx* ctx, artp_preprocessed* map_new, const artp_preprocessed* map_old,
                             float height_change_for_update, float* updated_out, int rect[4], uint64_t* n_updated);
/* Hole filling of a layer (rows x cols column-major floats, host): inpaintMatrix (art_planner/src/utils.cpp:13-64,
 * ARTP_INPAINT_PLANNER: NaN cells are the holes, cv::convertTo rounding, column 0 := column 1 / row 0 := row 1
 * afterwards) and the cost node's _elvMapProcess (cost_query_server.py:92-111, ARTP_INPAINT_COST_NODE: non-finite
 * cells are the holes, numpy truncation).  Like the reference the WHOLE layer comes back quantised to 8 bit over
 * [min, max] of its valid cells -- that part is restated exactly; the fill of the hole cells is by default a rim-inwards
 * distance-weighted mean (radius 3, on the device), with ARTP_INPAINT_TELEA Telea's marching like cv::inpaint (host).
 * OpenCV is not available here: both fills are unpinned.  A layer without holes is returned unchanged.
 * *n_holes (optional) = hole cells. */
enum { ARTP_INPAINT_PLANNER = 0, ARTP_INPAINT_COST_NODE = 1,
       /* OR-ed in: fill the holes by Telea's fast-marching method with cv::inpaint's conventions (radius 3; csrc/telea.h:
        * a restatement of the published algorithm, unpinned like everything OpenCV) on the image as the reference hands it
        * to cv::inpaint, instead of the rim-inwards mean.  Host code, ~10 ms for a 400 x 400 layer. */
       ARTP_INPAINT_TELEA = 2 };
int artp_inpaint_layer(artp_ctx* ctx, const float* layer, int rows, int cols, int mode, float* out, uint64_t* n_holes);
/* The fill alone (no context, no GPU): h x w row-major 8-bit image, mask != 0 = pixels to fill, radius `range`
 * (cv::inpaint(img, mask, out, range, cv::INPAINT_TELEA)); out may alias img.  h, w >= 2 (the march reads a 3 x 3
 * neighbourhood; a one-row / one-column layer in artp_inpaint_layer takes the rim-inwards fill instead). */
int artp_telea_inpaint_u8(const uint8_t* img, const uint8_t* mask, int h, int w, int range, uint8_t* out);
/* name: elevation, traversability, normal_x/_y/_z, plane_fit_std_dev, traversability_thresholded_no_safety,
 * traversability_thresholded, elevation_masked, sample_probability, cum_prob, observed, n_samples (the blurred
 * vertex density), traversability_sample_filter, updated (rows x cols floats each) or cum_prob_rowwise (rows). */
int artp_preprocessed_get_layer(artp_ctx* ctx, const artp_preprocessed* pp, const char* name, float* out);
/* Planner::setMap (planner.cpp:135-163): make the result the context's map -- both height fields with
 * their tables, the sampler layers and the z bounds.  When the context already holds a map of the same geometry, each
 * height field is compared with the installed one bit for bit first: an identical layer keeps its tables, a layer that
 * differs inside a rectangle of at most a quarter of the map is rewritten there only and its tables take the
 * rectangle-update path (artp_update_layer_rects) -- the same tables as a fresh install, at the cost of the change. */
int artp_preprocessed_install(artp_ctx* ctx, const artp_preprocessed* pp);
/* Map::reApplyPreprocessing (art_planner/src/map/map.cpp:94-96), the part that can change on an unchanged map:
 * the sampling distribution re-weighted by the inverse density of the given roadmap vertices (DEVICE pointer,
 * n x 7 doubles; NULL / 0 = no density term) and its CDF.  With install_sampler != 0 the context's sampler uses
 * the new CDF from the next sample on (the height fields are untouched). */
int artp_preprocessed_reweight_dev(artp_ctx* ctx, artp_preprocessed* pp, const struct artp_preprocess_params* params,
                                   const double* vertex_se3_dev, size_t n_vertices, int install_sampler);
void artp_preprocessed_destroy(artp_preprocessed* pp);

/* ---- learned motion cost: MotionCostObjective::MotionCostFunc
 *      (art_planner/include/art_planner/objectives/motion_cost_objective.h:22-23; the reference
 *      implements it as a ROS service to the Python/CUDA node, art_planner_ros/src/planner_ros.cpp:
 *      283-318 -> art_planner_motion_cost/scripts/cost_query_server.py:145-169) ----------------------
 * Weights: a flat float32 blob of the n48convNetwork3LR network (network_light.py:19-62) with eval-mode
 * BatchNorm folded into every convolution: "ARMC", version byte 1, 3 pad bytes, then for the six
 * convolutions [Cout][Cin][KH][KW] weights + [Cout] bias, then the 1x1 layers tar0, out0, out1_conv1..3
 * ([Cout][Cin] + [Cout]) and out2_conv1..3 ([Cin] + 1).  tools/convert_weights.py writes it from a
 * torch state_dict. */
size_t artp_cost_blob_bytes(void);
int artp_cost_load_weights(artp_ctx* ctx, const void* blob, size_t bytes);
/* Which kernel answers artp_cost_query: *mfma = 1 the MFMA form of FCpart (network_light.py:113-165), 0 the fp32 VALU kernels.
 * artp_cost_load_weights runs a probe batch through both and falls back to fp32 if they disagree (*selfcheck: 1 agreed,
 * 0 disagreed, -1 not run; *max_abs_diff = the probe's largest difference).  Any pointer may be NULL. */
int artp_cost_fc_path(artp_ctx* ctx, int* mfma, int* selfcheck, float* max_abs_diff);
/* mfma == 0: artp_cost_query on the fp32 VALU kernels (comparison / a toolchain whose self-check fails); != 0 (default): the
 * MFMA form, entered only through the self-check.  Before artp_cost_load_weights it sets what the load will choose. */
int artp_cost_set_fc_path(artp_ctx* ctx, int mfma);
/* CostPredictor.updateFeatures (predictor.py:28-36) + CostQuery.setMapParams (cost_query.py:26-35):
 * elev_xy is the server's map array [rows][cols] row-major with index a growing along world x and b
 * along world y (cost_query_server.py:66-74), holes already inpainted; (cx, cy) = map centre. */
int artp_cost_update_map(artp_ctx* ctx, const float* elev_xy, int rows, int cols, double res, double len_x,
                         double len_y, double cx, double cy);
/* The same with the map array already in HBM (asynchronous on the context's stream). */
int artp_cost_update_map_dev(artp_ctx* ctx, const float* elev_xy_dev, int rows, int cols, double res, double len_x,
                             double len_y, double cx, double cy);
/* The same from the planner's grid_map elevation layer (column-major rows x cols, as PlannerRos publishes it
 * to the cost node): applies the server's rot90(.., 2).transpose() re-indexing (cost_query_server.py:66-74).
 * Holes (NaN / inf) are an error unless artp_cost_set_hole_filling is on (cost_query_server.py:90-111). */
int artp_cost_update_map_layer(artp_ctx* ctx, const float* layer, int rows, int cols, double res, double len_x,
                               double len_y, double pos_x, double pos_y);
/* enabled != 0: artp_cost_update_map_layer fills the holes of its layer itself (artp_inpaint_layer,
 * ARTP_INPAINT_COST_NODE) instead of rejecting it -- the node's behaviour (cost_query_server.py:92-111).
 * enabled == 2: with Telea's fill (ARTP_INPAINT_TELEA), the algorithm the node calls. */
int artp_cost_set_hole_filling(artp_ctx* ctx, int enabled);
/* MotionCostFunc: edges [B][6] = target x y yaw, start x y yaw (prm_motion_cost.cpp:41-52);
 * cost [B][3] = energy, time, risk (= 1 - prob; cost_query.py:65-69). */
int artp_cost_query(artp_ctx* ctx, const float* edges, size_t b, float* cost);
int artp_cost_query_dev(artp_ctx* ctx, const float* edges, size_t b, float* cost);
/* The MotionCostFunc seam of PRMMotionCostMaintainer (prm_motion_cost.cpp:27-73: `func(edge_matrix, &edge_cost)` -- in
 * PlannerRos a ROS service client, planner_ros.cpp:283-318).  With fn != NULL every batch the ROADMAP prices with the
 * learned objective (artp_roadmap_params::objective == 2: build, grow, revalidate, set_query, simplify_path) goes
 * through the caller's function instead of the device network: edges / cost are HOST buffers in artp_cost_query's
 * layout, return 0 on success; any other value fails the roadmap call with ARTP_ERR_COST_FUNC (the reference throws
 * std::runtime_error("Motion cost call failed"), motion_cost_objective.cpp:78-83).  No weights need to be loaded then.
 * fn == NULL (default): device pricing (artp_cost_query_dev).  artp_cost_query itself is never redirected. */
typedef int (*artp_cost_query_fn)(void* user, const float* edges, size_t b, float* cost);
int artp_cost_set_external_query(artp_ctx* ctx, artp_cost_query_fn fn, void* user);
/* diagnostics: the feature-map cell (row, col) CostQuery.__call__ gathers for each edge's start position
 * (cost_query.py:54-55: float64 arithmetic, clamp to [1, shape - 2], .long()) -- computed by the device function the
 * cost kernels use; the tests compare it with the reference's own CostQuery.  Host buffers. */
int artp_cost_debug_query_cells(artp_ctx* ctx, const float* edges, size_t b, int32_t* rows_out, int32_t* cols_out);
/* diagnostics: feature map as float [fh][fw][48]; out may be NULL to query the size */
int artp_cost_get_features(artp_ctx* ctx, float* out, int* fh, int* fw);

#ifdef __cplusplus
}
#endif
#endif /* ARTP_C_H */
