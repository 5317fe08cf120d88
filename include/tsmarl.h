/*
 * tsmarl.h -- C-ABI of the MI355X-native multi-agent rollout + update hot path.
 *
 * Drop-in boundary for the data-parallel hot path of eric-downes/tianshou_marl
 * (BASELINE.json north_star; scope table SURVEY.md section 8).  The reference is pure Python:
 * its only "native" entry points on this path are the numba @njit kernels and the numpy/torch
 * calls listed next to each function below (paths relative to /root/reference).  A maintainer
 * binds these symbols with ctypes (INTEGRATION.md shows the stubs) in place of those call sites.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types.
 *  - Every data pointer is a DEVICE (HBM) pointer unless its name ends in `_host`.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous
 *    on that stream; nothing here allocates, frees or synchronises (graph-capture safe) except
 *    the tsm_mem_* / tsm_stream_sync helpers.
 *  - Return value: 0 = TSM_OK, otherwise a TSM_ERR_* code; tsm_last_error() returns a
 *    thread-local message.  The Python host maps TSM_ERR_INVALID -> ValueError,
 *    TSM_ERR_MALFORMED_BUFFER -> MalformedBufferError (tianshou/data/buffer/buffer_base.py:380),
 *    others -> RuntimeError.
 *  - Device layout ("joint-step lanes"): time-major SoA
 *      field[slot][env][agent][...]   slot in [0, sub_size), env in [0, buffer_num)
 *    so that the (env, agent) lane index is the fastest-varying dimension of every per-step scalar.
 *    Reference flat buffer index  <->  (env, slot):  index = env * sub_size + slot
 *    (tianshou/data/buffer/manager.py:38-46, vecbuf.py:35).
 *  - dtypes: payload f32, actions i32, flags u8 (0/1), indices/counters i64, episode-return
 *    accumulators f64 (the reference keeps rew/ep_return in f64, buffer_base.py:377,484).
 */
#ifndef TSMARL_H
#define TSMARL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSM_ABI_VERSION 4  /* 4: tsm_rollout_tag_desc.vnext_store, TSM_MAX_GATHER_FIELDS 12, tsm_random_permutations_advance; 3: tsm_slab_seg.frag_image, tsm_critic_rows_grad_*(w1_image), tsm_kernel_option_* */

enum {
    TSM_OK = 0,
    TSM_ERR_INVALID = 1,          /* bad argument (ValueError) */
    TSM_ERR_HIP = 2,              /* HIP runtime / launch failure */
    TSM_ERR_MALFORMED_BUFFER = 3, /* buffer_base.py:380-386 */
    TSM_ERR_UNSUPPORTED = 4
};

int tsm_abi_version(void);
const char *tsm_last_error(void);
/* name_out: >= 64 bytes.  Fails with TSM_ERR_HIP when no gfx950 device is visible. */
int tsm_device_info(int *n_cu, int *wave_size, int64_t *hbm_bytes, char *name_out);

/* Kernel selection options.  Where one entry point has two kernels behind it the choice is a rule over the problem size; an
 * option overrides the rule for this process.  Default: the environment variable (read at first use); tsm_kernel_option_set
 * replaces it at any time (hosts that cache launches -- hipGraphs -- key them by the option values).  No reference counterpart.
 *   "actor_tile"   0 = by minibatch size | 32 | 64 (tsm_ppo_actor_rows_update / _grid)            TSM_ACTOR_TILE
 *   "split_bf16"   0 | 1 = layer 1 of tsm_critic_rows_forward on the bf16 matrix pipe, three-way split operands, f32
 *                  accumulation (experimental, never the default)                                   TSM_SPLIT_BF16
 *   "rollout_form" tsm_rollout_spread / tsm_rollout_spread_actor: 0 = by rule | 1 = tile form | 2 = wave-autonomous form   TSM_ROLLOUT_FORM */
int tsm_kernel_option_get(const char *name, int32_t *value_out);
int tsm_kernel_option_set(const char *name, int32_t value);

/* memory / stream helpers for hosts that do not bring their own allocator (PyTorch does) */
int tsm_mem_alloc(void **dptr, int64_t bytes);
int tsm_mem_free(void *dptr);
int tsm_mem_h2d(void *dst, const void *src_host, int64_t bytes, void *stream);
int tsm_mem_d2h(void *dst_host, const void *src, int64_t bytes, void *stream);
int tsm_mem_set(void *dst, int value, int64_t bytes, void *stream);
int tsm_stream_sync(void *stream);
/* Ends a hipGraph capture that failed half-way on `stream` (drops the partial graph); returns 1 if one was ended, else 0. */
int tsm_stream_abort_capture(void *stream);

/* ---------------------------------------------------------------------------------------------
 * GAE  [SURVEY 8a: a11, a12]
 * Replaces  Algorithm.compute_episodic_return + value_mask + numba `_gae`
 *           (tianshou/algorithm/algorithm_base.py:631-717, 1079-1134)
 *           and the return_scaling arithmetic of a2c.py:132-146.
 * One independent series per lane; inputs/outputs are [T][n_lane] time-major.
 *   v_next' = v_s_next * v_scale * !terminated;  v' = v_s * v_scale
 *   end     = terminated | truncated | (row is the lane's last row)      (unfinished_index forcing)
 *   delta = rew + gamma*v_next' - v';  adv_t = delta_t + gamma*lambda*(1-end_t)*adv_{t+1}  (f64)
 *   adv_out = f32(adv);  returns_out = f32((adv + v') / v_scale)
 * env_len / env_start (nullable, [n_lane / lanes_per_env] i32): ragged / rotated sub-buffers --
 *   lane rows are slots (start + k) % T for k in [0, len); rows outside are left untouched.
 * terminated / truncated: u8 [T][n_lane] when flags_per_lane != 0, else [T][n_lane/lanes_per_env].
 * ------------------------------------------------------------------------------------------- */
int tsm_gae_lanes(const float *v_s, const float *v_s_next, const float *rew,
                  const uint8_t *terminated, const uint8_t *truncated, int flags_per_lane,
                  int64_t T, int64_t n_lane, int64_t lanes_per_env, const int32_t *env_start,
                  const int32_t *env_len, double gamma, double gae_lambda, double v_scale,
                  float *returns_out, float *adv_out, void *stream);
/* The same with `return_scaling` statistics that live in HBM (a2c.py:132-146): rms = f64 {mean, var, count} of
 * the reference's RunningMeanStd (utils/statistics.py:68-114); v_scale = sqrt(rms[1] + rms_eps) is read by the
 * kernel, so a captured hipGraph follows the statistics from update to update. */
int tsm_gae_lanes_rms(const float *v_s, const float *v_s_next, const float *rew,
                      const uint8_t *terminated, const uint8_t *truncated, int flags_per_lane,
                      int64_t T, int64_t n_lane, int64_t lanes_per_env, const int32_t *env_start,
                      const int32_t *env_len, double gamma, double gae_lambda, const double *rms,
                      double rms_eps, float *returns_out, float *adv_out, void *stream);
/* The few-lanes / long-series form of tsm_gae_lanes (n_lane <= 64, T >= 1024: the MARL trainers' one time-ordered lane of
 * n_env * T rows, training_coordinator.py:118,154,336) runs its super-chunks of 4096 steps on different workgroups when the
 * host has registered a ZEROED device workspace of tsm_gae_scan_workspace_bytes() bytes for the CURRENT device (one per device;
 * the memory stays the caller's; nullptr withdraws it): same bits as the one-workgroup-per-lane form, 12 800 rows 27 -> 9 us.
 * The workspace is cut into slots -- one per eager stream, one per launch recorded into a hipGraph -- so launches in flight never
 * share hand-over state; without a free slot, or on a device without a workspace, the one-workgroup form runs.
 * tsm_gae_set_scan_error_word: a word of PINNED host memory that a scan whose bounded wait ran out sets to 1 (its outputs are NaN
 * from there on); the host reads it where it reads its statistics. */
int64_t tsm_gae_scan_workspace_bytes(void);
int tsm_gae_set_scan_workspace(void *workspace, int64_t bytes);
int tsm_gae_set_scan_error_word(int32_t *host_pinned);

/* RunningMeanStd.update (utils/statistics.py:97-114) with the UNNORMALISED returns of a2c.py:144-146:
 * x[i] = returns[ids ? ids[i] : i] * sqrt(rms[1] + rms_eps), i < n; batch mean / population variance in f64
 * (two-level fixed-order sums), then the parallel-variance merge into rms = {mean, var, count} in place.
 * work: f64[tsm_rms_update_work_elems(n)]. */
int64_t tsm_rms_update_work_elems(int64_t n);
int tsm_rms_update(const float *returns, const int64_t *ids, int64_t n, double *rms, double rms_eps,
                   double *work, void *stream);

/* `episode_mc_return_to_go` (algorithm_base.py:1137-1151) over n_lane independent episodes
 * stored [T][n_lane]; out f32. */
int tsm_mc_return_to_go_lanes(const float *rew, int64_t T, int64_t n_lane, double gamma,
                              float *out, void *stream);

/* Several row-gathers with element conversion in ONE launch: field k copies n_rows rows of `width` elements,
 *   dst[r][c] = convert(src[src_row(r) * src_row_stride + src_offset + c]),
 * src_row(r) = r (T == 0) or, with T > 0 and n_rows == T * E, the time-major slot of env-major row r = e * T + t:
 * src_row = t * E + e -- the reference's flat index order (manager.py:131-193) read out of a [T][E][...] store; src_offset
 * picks an agent's column of a joint row.  Kinds: f32 -> f32; i32 / i64 / u8 -> i32 / i64 / f32 / u8 (u8: 0 / 1, torch.bool's
 * storage).  Replaces the per-field torch copies of the MARL trainers' per-agent batches (training_coordinator.py:118,154,336) and
 * of learn()'s static buffers.  No reference counterpart. */
#define TSM_MAX_GATHER_FIELDS 12
enum { TSM_KIND_F32 = 0, TSM_KIND_I32 = 1, TSM_KIND_I64 = 2, TSM_KIND_U8 = 3 };
typedef struct tsm_gather_field {
    const void *src;
    void *dst;
    int64_t n_rows;
    int64_t T, E;
    int64_t src_row_stride, src_offset;   /* in elements of the source type */
    int32_t width, src_kind, dst_kind, _pad;
} tsm_gather_field;
int tsm_gather_fields(const tsm_gather_field *fields_host, int32_t n_fields, void *stream);

/* V(obs_next) (a2c.py:124) without a second critic pass over every row.  Rows written by a Collector are chained:
 * obs_next of slot t is obs of slot t + 1 unless the episode ended at t (collector.py:1040-1069), so for T unrotated,
 * equally filled slots of U units (lanes, or joint rows for a centralized critic)
 *   v_next[t][u] = v_s[t + 1][u] (t < T - 1),  v_last[u] = V(obs_next of the last slot) (t = T - 1)
 * provided no episode ended before the last slot: *flag == 0, flag = tsm_any_nonzero_u8(done[0 .. T-1)).  If one did,
 * v_next = v_full, the complete pass (run it with tsm_mlp_forward_cond on the same flag).  Bit-identical to the full
 * pass either way (the critic's rows are computed independently of each other). */
int tsm_any_nonzero_u8(const uint8_t *x, int64_t n, int32_t *flag_out, void *stream);
int tsm_value_next_select(const float *v_s, const float *v_last, const float *v_full, const int32_t *flag, int64_t T,
                          int64_t U, float *v_next_out, void *stream);
/* The same for ENV-major rows [E][T][U] (flat reference index order, sample_indices(0): what the MARL trainers' per-agent
 * batches hold): v_next[e][t][u] = v_s[e][t + 1][u] (t < T - 1), v_last[e][u] (t = T - 1), or v_full when *flag != 0. */
int tsm_value_next_select_env_major(const float *v_s, const float *v_last, const float *v_full, const int32_t *flag,
                                    int64_t E, int64_t T, int64_t U, float *v_next_out, void *stream);
/* V(obs_next) for a buffer that does not store obs_next (ReplayBuffer(ignore_obs_next=True), buffer_base.py:612-616: obs_next
 * is read as obs[next(index)], and next(index) is the row itself at an episode end and at the newest row): for T unrotated,
 * equally filled slots v_next[t][u] = (t == T - 1 || done[t][u / lanes_per_env]) ? v_s[t][u] : v_s[t + 1][u].
 * v_s, v_next_out f32 [T][U]; done u8 [T][U / lanes_per_env]. */
int tsm_value_next_index(const float *v_s, const uint8_t *done, int64_t T, int64_t U, int64_t lanes_per_env,
                         float *v_next_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * VectorReplayBuffer  [a8, a9]
 * Replaces  ReplayBufferManager.add / _update_state_pre_add / sample_indices(0) /
 *           unfinished_index / _prev_index / _next_index / reset
 *           (tianshou/data/buffer/manager.py:70-229,306-358; buffer_base.py:292-410,495-531;
 *            vecbuf.py:33-37).
 * `state` is one device allocation of tsm_vrb_state_bytes() bytes owned by the caller:
 *   i64 insertion_idx[B], size[B], ep_len[B], ep_start_idx[B], last_index[B], lengths[B];
 *   f64 ep_return[B][rew_dim];  i64 error_flag[1]
 * ------------------------------------------------------------------------------------------- */
int64_t tsm_vrb_state_bytes(int64_t buffer_num, int64_t rew_dim);
int tsm_vrb_init(void *state, int64_t buffer_num, int64_t sub_size, int64_t rew_dim, void *stream);
int tsm_vrb_reset(void *state, int64_t buffer_num, int64_t sub_size, int64_t rew_dim,
                  int keep_statistics, void *stream);

/* One payload field scattered by tsm_vrb_add: row r of `src` ([R][row_bytes], the vector env's
 * AoS step output) goes to `dst` + ((slot * buffer_num + env) * row_bytes)  (time-major SoA). */
typedef struct {
    const void *src;
    void *dst;
    int64_t row_bytes;
} tsm_field;

/* add(): R rows, row r belongs to sub-buffer buffer_ids[r] (NULL = arange(R)); ids must be unique
 * within one call (the reference's Collector never repeats an env id within a step).
 *   rew [R][rew_dim] f32, done [R] u8 (= terminated | truncated, manager.py:150).
 *   done_store: u8 [sub_size][buffer_num] (read by prev/next/unfinished_index).
 * Outputs [R]: ptr (flat reference index env*sub_size+slot), ep_rew [R][rew_dim] f64, ep_len,
 * ep_idx -- the reference's 4-tuple (manager.py:193).  MalformedBufferError is reported through
 * state.error_flag (checked by tsm_vrb_check). */
int tsm_vrb_add(void *state, int64_t buffer_num, int64_t sub_size, int64_t rew_dim,
                const int64_t *buffer_ids, int64_t R, const float *rew, const uint8_t *done,
                uint8_t *done_store, const tsm_field *fields_host, int n_fields, int64_t *ptr_out,
                double *ep_rew_out, int64_t *ep_len_out, int64_t *ep_idx_out, void *stream);
/* synchronises `stream`; returns TSM_ERR_MALFORMED_BUFFER if any add() tripped buffer_base.py:380 */
int tsm_vrb_check(void *state, int64_t buffer_num, int64_t rew_dim, void *stream);

/* sample_indices(0): out [buffer_num*sub_size] i64 (first *n_out valid), env-major, time-ordered.
 * n_out: device i64[1].  scratch: device i64[buffer_num + 1]. */
int tsm_vrb_sample_indices_all(const void *state, int64_t buffer_num, int64_t sub_size,
                               int64_t *out, int64_t *n_out, int64_t *scratch, void *stream);
/* unfinished_index(): out [buffer_num] i64 (first *n_out valid, ascending env order). */
int tsm_vrb_unfinished_index(const void *state, int64_t buffer_num, int64_t sub_size,
                             const uint8_t *done_store, int64_t *out, int64_t *n_out, void *stream);
int tsm_vrb_prev(const void *state, int64_t buffer_num, int64_t sub_size, const uint8_t *done_store,
                 const int64_t *index, int64_t n, int64_t *out, void *stream);
int tsm_vrb_next(const void *state, int64_t buffer_num, int64_t sub_size, const uint8_t *done_store,
                 const int64_t *index, int64_t n, int64_t *out, void *stream);
/* gather rows by flat reference index (ReplayBuffer.__getitem__, buffer_base.py:591-635):
 * out[i] = store[(slot*buffer_num + env)] for index[i] = env*sub_size + slot. */
int tsm_vrb_gather(const void *store, int64_t buffer_num, int64_t sub_size, int64_t row_bytes,
                   const int64_t *index, int64_t n, void *out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Agent dispatch  [a5, a10]
 * Replaces  `np.nonzero(batch.obs.agent_id == agent_id)[0]` per agent and the scatter
 *           `holder.act[agent_index] = act`  (tianshou/algorithm/multiagent/marl.py:148,170-180,233).
 * agent_id [B] i32 in [0, n_agent).  index_out [B] i64: rows of agent 0 (ascending), then agent 1 ...
 * offsets_out [n_agent+1] i64.  scratch: i64 [n_agent * n_blocks + 1], n_blocks = ceil(B/1024).
 * ------------------------------------------------------------------------------------------- */
int tsm_agent_index(const int32_t *agent_id, int64_t B, int32_t n_agent, int64_t *index_out,
                    int64_t *offsets_out, int64_t *scratch, void *stream);
/* dst[index[i]] = src[i]  /  dst[i] = src[index[i]]   rows of row_bytes */
int tsm_scatter_rows(const void *src, const int64_t *index, int64_t n, int64_t row_bytes, void *dst,
                     void *stream);
int tsm_gather_rows(const void *src, const int64_t *index, int64_t n, int64_t row_bytes, void *dst,
                    void *stream);

/* ---------------------------------------------------------------------------------------------
 * Categorical policy head  [a7, a13]
 * Replaces  torch.distributions.Categorical(logits=...).sample / log_prob / entropy
 *           (tianshou/utils/net/discrete.py:22-24; reinforce.py:183-189; ppo.py:160,187,210).
 * logits [B][A] f32 row-major.  Sampling uses a counter-based Philox4x32-10 stream keyed by
 * (seed, offset + row): reproducible, order-independent, NOT bit-identical to torch's CPU RNG.
 * deterministic != 0 -> dist.mode (argmax) (reinforce.py:185-189).  offset_dev (nullable, device u64[1]) is
 * added to `offset`, so that a captured hipGraph advances the stream from device memory.
 * ------------------------------------------------------------------------------------------- */
int tsm_categorical_sample(const float *logits, int64_t B, int32_t A, uint64_t seed,
                           uint64_t offset, const uint64_t *offset_dev, int deterministic, int32_t *act_out,
                           float *logp_out, void *stream);
int tsm_categorical_logp_entropy(const float *logits, const int32_t *act, int64_t B, int32_t A,
                                 float *logp_out, float *ent_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * PPO clip loss, forward + backward w.r.t. logits and value  [a14]
 * Replaces  the body of PPO._update_with_batch (tianshou/algorithm/modelfree/ppo.py:182-211)
 *           between the network forward and optim.step.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double eps_clip;    /* 0.2 */
    double dual_clip;   /* <= 0: off */
    double vf_coef;     /* 0.5 */
    double ent_coef;    /* 0.01 */
    int32_t value_clip; /* 0 */
    int32_t adv_norm;   /* 1 */
    /* 0: PPO clip objective (ppo.py:182-211).  1: plain policy gradient, actor_loss = -mean(log_prob * adv), the
     * actor term of A2C (a2c.py:260-270: + vf_coef * mse(returns, value) - ent_coef * entropy, no clipping) and, with
     * vf_coef = ent_coef = 0 and adv = returns, the whole loss of Reinforce (reinforce.py:373-379).  eps_clip,
     * dual_clip, value_clip and logp_old are ignored for kind 1.
     * 2 (tsm_ppo_loss_fwd_bwd only): the VALUE term alone -- vf_coef * value loss, d loss / d value; logits, act,
     * logp_old, adv, adv_stats and dlogits_out may be NULL (the policy terms of the same samples come from
     * tsm_ppo_actor_rows_update). */
    int32_t loss_kind;
    /* <= 1: one critic value per sample.  N > 1 (tsm_ppo_loss_fwd_bwd only): centralized critic -- samples
     * i = r * N + a of a minibatch are the N agents of joint row r, `value` holds one entry per ROW (value[i / N]) and
     * dvalue_out [M / N] receives the sum of the row's N per-sample gradients (M % N == 0). */
    int32_t value_group;
} tsm_ppo_cfg;

/* Per-minibatch advantage statistics (ppo.py:185, torch unbiased std).  Minibatch k covers
 * perm[mb_start[k] .. mb_start[k+1]) (perm == NULL: identity; mb_start: device i64[n_mb+1]).
 * stats_out [n_mb][2] f32 = {mean, std}.  One launch serves every minibatch of an epoch. */
int tsm_ppo_adv_stats(const float *adv, const int64_t *perm, const int64_t *mb_start,
                      int32_t n_mb, float *stats_out, void *stream);
/* The same statistics for long minibatches: every 8192-row chunk of a minibatch is reduced by its own workgroup
 * (shifted f64 sums), a second launch folds the chunks in chunk order (deterministic; independent of the grid).
 * max_rows >= the longest minibatch (host knowledge: mb_start lives in HBM); work: f64[tsm_ppo_adv_stats_work_elems]. */
int64_t tsm_ppo_adv_stats_work_elems(int32_t n_mb, int64_t max_rows);
int tsm_ppo_adv_stats_wide(const float *adv, const int64_t *perm, const int64_t *mb_start, int32_t n_mb,
                           int64_t max_rows, double *work, float *stats_out, void *stream);

/* Env-sharded replicas: the advantage normalisation of ppo.py:184-186 over the GLOBAL minibatch (the union
 * of the ranks' parts).  pack: stats [n_mb][2] = this rank's {mean, unbiased std} and its row counts mb_start[k+1] -
 * mb_start[k] -> pack_out f64 [n_mb][3] = {n, sum x, sum x^2}; the caller sums pack over the ranks (one all-reduce
 * for every minibatch of the update); unpack: summed pack -> stats_out [n_mb][2] = {mean, unbiased std} of the union. */
int tsm_ppo_adv_stats_pack(const float *stats, const int64_t *mb_start, int32_t n_mb, double *pack_out, void *stream);
int tsm_ppo_adv_stats_unpack(const double *pack, int32_t n_mb, float *stats_out, void *stream);

/* One minibatch of M samples: sample i is row perm[i] of the full-batch arrays (perm == NULL:
 * row first_row + i).  logits [M][A] and value [M] are minibatch-contiguous network outputs.
 * Outputs: dlogits [M][A], dvalue [M] (already scaled by 1/M and the coefficients, i.e. d loss),
 * partial [n_blocks(M)][4] f64 per-block sums {clip_obj, vf, ent, 0}; tsm_ppo_loss_finalize
 * folds them into scalars_out[4] = {loss, clip_loss, vf_loss, ent_loss}. */
int64_t tsm_ppo_loss_partial_elems(int64_t M);
int tsm_ppo_loss_fwd_bwd(const float *logits, const float *value, const int32_t *act,
                         const float *logp_old, const float *adv, const float *returns,
                         const float *v_s_old, const int64_t *perm, int64_t first_row, int64_t M,
                         int32_t A, const float *adv_stats, const tsm_ppo_cfg *cfg_host,
                         float *dlogits_out, float *dvalue_out, double *partial_out, void *stream);
int tsm_ppo_loss_finalize(const double *partial, int64_t M, const tsm_ppo_cfg *cfg_host,
                          float *scalars_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step  [a15]
 * Replaces  Algorithm.Optimizer.step: clip_grad_norm_ + torch.optim.Adam.step
 *           (tianshou/algorithm/algorithm_base.py:485-498; optim.py:91-111) on one flat f32
 *           parameter vector (actor+critic union, utils/net/common.py:461-474).
 * grad_slabs [n_slab][n] f32: per-workgroup partial gradients, summed here in slab order
 * (deterministic).  max_grad_norm <= 0: no clipping; otherwise work: f32[tsm_adam_work_elems(n)] device.
 * param_image / image_map (nullable): padded LDS-layout copy of the parameters kept in sync for the fused
 * MLP kernels (tsm_policy_image_elems / tsm_policy_image_map).
 * step: 1-based Adam step count; when step_dev (device i64[1]) is given it overrides `step`.
 * lr_dev (nullable, device f64[1]) overrides `lr`: the learning rate an LR scheduler steps after every update
 * (algorithm_base.py:626-627, optim.py:22-46) lives in HBM, so a captured hipGraph follows the schedule.
 * ------------------------------------------------------------------------------------------- */
/* out[i] = scale * sum_s grad_slabs[s][i]: the flat gradient handed to the RCCL all-reduce of the
 * env-sharded data-parallel path (no reference counterpart: the reference has no distributed backend). */
int tsm_reduce_slabs(const float *grad_slabs, int32_t n_slab, int64_t n, double scale, float *out,
                     void *stream);
int64_t tsm_adam_work_elems(int64_t n);
int tsm_adam_step(float *param, const float *grad_slabs, int32_t n_slab, int64_t n, float *exp_avg,
                  float *exp_avg_sq, int64_t step, const int64_t *step_dev, double lr, const double *lr_dev,
                  double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, float *work,
                  float *param_image, const int32_t *image_map, void *stream);
/* Segmented forms: the flat vector is covered, in order, by up to TSM_MAX_SLAB_SEGS segments whose gradients come from
 * different kernels (tsm_ppo_actor_rows_update / tsm_ppo_critic_rows_update write separate slab arrays): the gradient of
 * parameter offset + i, i < n, in slab s is slabs[s * stride + i].  One launch for the whole vector; with
 * max_grad_norm > 0 one reduction launch in front and ONE norm over all segments (clip_grad_norm_ over
 * ActorCritic.parameters(), algorithm_base.py:485-498). */
#define TSM_MAX_SLAB_SEGS 6
typedef struct tsm_slab_seg {
    const float *slabs;
    int64_t offset, n, stride;
    int32_t n_slab;
    int32_t frag_k1;        /* with frag_image: the segment is a row-major [128][frag_k1] first-layer weight matrix */
    const float *scale_dev; /* nullable device f32[1]: the segment's summed gradient is multiplied by it (a loss whose
                               gradient is a device-side scalar times a sum, e.g. CTDEPolicy's actor loss, ctde.py:185) */
    float *frag_image;      /* nullable: the updated parameters are ALSO stored in the fragment order the one-launch critic
                               kernels load with coalesced 16-B loads (tsm_critic_rows_w1_image: same layout), so the
                               next gradient step need not gather W1 through 64-B pieces of 1.5 KB rows */
    int32_t frag_kj, _pad;  /* k-groups of 16 per image row: tsm_critic_rows_w1_image_kj(frag_k1) */
} tsm_slab_seg;
int tsm_reduce_slabs_segs(const tsm_slab_seg *segs_host, int32_t n_seg, int64_t n, double scale, float *out,
                          void *stream);
int tsm_adam_step_segs(float *param, const tsm_slab_seg *segs_host, int32_t n_seg, int64_t n, float *exp_avg,
                       float *exp_avg_sq, int64_t step, const int64_t *step_dev, double lr, const double *lr_dev,
                       double beta1, double beta2, double eps, double weight_decay, double max_grad_norm, float *work,
                       void *stream);
/* param_image[image_map[i]] = param[i]  (initial fill of the padded image; pads must already be zero) */
int tsm_scatter_image(const float *param, int64_t n, const int32_t *image_map, float *param_image,
                      void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused actor + critic MLP (f32 MFMA)  [a7, a11, a13, a14]
 * Networks: obs[D] -> Linear(H) ReLU -> Linear(H) ReLU -> {Linear(A) logits | Linear(1) value},
 * i.e. Net(hidden_sizes=[H,H]) + DiscreteActor(softmax_output=False) / DiscreteCritic with separate
 * trunks (tianshou/utils/net/common.py:246-369, discrete.py:27-124).  `params` is ONE flat f32 vector
 * in ActorCritic.parameters() order (common.py:461-474):
 *   actor W1[H][D] b1[H] W2[H][H] b2[H] W3[A][H] b3[A] | critic W1[H][D] b1[H] W2[H][H] b2[H] W3[1][H] b3[1]
 * Supported: H == 64, D <= 64, A <= 16 (others: TSM_ERR_INVALID).
 *
 * tsm_policy_forward replaces ProbabilisticActorPolicy.forward + critic(obs) + dist.log_prob
 * (reinforce.py:167-192, a2c.py:121-127, ppo.py:157-161) for obs [B][D]:
 *   mode 0: logits/value only;  1: sample (Philox, key=seed, counter=offset+row);  2: dist.mode;
 *   mode 3: log-prob of the GIVEN actions act_io.   logits_out/value_out/logp_out are nullable.
 *   offset_dev (nullable, device u64[1]) is added to `offset`: lets a captured hipGraph advance the
 *   sampling counter from device memory (tsm_mpe_spread_step bumps it).
 *
 * tsm_ppo_update_fused replaces one gradient step of PPO._update_with_batch (ppo.py:182-212) up to the
 * parameter gradients: network forward, loss (as tsm_ppo_loss_fwd_bwd) and the whole backward pass.
 * Sample i of the minibatch is row perm[i] (perm == NULL: first_row + i) of obs [n][D], act, logp_old,
 * adv, returns, v_s_old.  Each of the n_blocks workgroups writes one gradient slab:
 * grad_slabs_out [n_blocks][P]; feed them to tsm_adam_step(n_slab = n_blocks).
 * loss_partial_out: f64 [n_blocks][4]; scalars_out (nullable) f32[4] = {loss, clip, vf, ent}.
 * opt_step_dev (nullable, device i64[1]): incremented by one per call -- the device-resident optimizer
 * step count that tsm_adam_step(step_dev=...) reads, so a captured hipGraph can be replayed.
 * ------------------------------------------------------------------------------------------- */
int64_t tsm_policy_param_count(int32_t obs_dim, int32_t hidden, int32_t n_act);
/* Padded parameter image: the exact LDS layout of both nets (zero pads included).  When `param_image` is
 * given to the kernels below they stage it with straight 16-B copies instead of re-packing `params`. */
int64_t tsm_policy_image_elems(int32_t obs_dim, int32_t hidden, int32_t n_act);
int tsm_policy_image_map(int32_t obs_dim, int32_t hidden, int32_t n_act, int32_t *map_out_host);
int tsm_policy_forward(const float *params, const float *param_image, int32_t obs_dim, int32_t hidden,
                       int32_t n_act,
                       const float *obs, int64_t B, int mode, uint64_t seed, uint64_t offset,
                       const uint64_t *offset_dev, float *logits_out, float *value_out, int32_t *act_io, float *logp_out,
                       void *stream);
/* recommended n_blocks (= number of gradient slabs; the launch is n_blocks x 2 workgroups, one net each) for a
 * minibatch of M rows.  NOT monotone in M (a slab is 45 KB written here and read back by tsm_adam_step, so past a full
 * chip fewer slabs with two tiles each are faster): when one workspace serves minibatches of several sizes, size
 * grad_slabs_out / loss_partial_out by the LARGEST value over those sizes, not by the value at the largest size.
 * tsm_ppo_update_fused writes n_blocks * n_param floats and n_blocks * 4 doubles. */
int tsm_ppo_update_grid(int64_t M, int32_t max_blocks);
int tsm_ppo_update_fused(const float *params, const float *param_image, int32_t obs_dim, int32_t hidden,
                         int32_t n_act,
                         const float *obs, const int32_t *act, const float *logp_old, const float *adv,
                         const float *returns, const float *v_s_old, const int64_t *perm,
                         int64_t first_row, int64_t M, const float *adv_stats,
                         const tsm_ppo_cfg *cfg_host, int32_t n_blocks, float *grad_slabs_out,
                         double *loss_partial_out, float *scalars_out, int64_t *opt_step_dev,
                         void *stream);

/* Loss statistics of many gradient steps in ONE launch (pass scalars_out = NULL to tsm_ppo_update_fused):
 * step k reads loss partials at partial + k*stride_elems (n_blocks_dev[k] rows of 4 f64) and M_dev[k];
 * scalars_out f32 [n_steps][4] = {loss, clip_loss, vf_loss, ent_loss} (the 4 .item() of ppo.py:213-216). */
int tsm_ppo_finalize_many(const double *partial, int64_t stride_elems, const int32_t *n_blocks_dev,
                          const int64_t *M_dev, int32_t n_steps, const tsm_ppo_cfg *cfg_host,
                          float *scalars_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batched MPE worlds  [SURVEY 8f-1; replaces the per-env Python loop of a1-a3 for simple_spread]
 * Device-resident restatement of PettingZoo-MPE simple_spread behind the parallel-mode
 * EnhancedPettingZooEnv data formats (tianshou/env/enhanced_pettingzoo_env.py:130-222) and the
 * BaseVectorEnv.step/reset contract (tianshou/env/venvs.py:195-322).  pettingzoo's sources are not in
 * the reference tree: dynamics parity is UNPINNED.  State (caller-owned, device):
 *   agent_pos, agent_vel, landmark_pos f32 [n_env][N][2]; steps i32 [n_env]; episode_ctr u64 [n_env].
 * obs layout per agent (6N): [vel(2), pos(2), landmarks rel (2N), others rel (2(N-1)), comm (2(N-1))].
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_env, n_agent, max_cycles, _pad;
    double dt, damping, contact_force, contact_margin, agent_size, landmark_size, accel, max_speed,
        local_ratio;
} tsm_mpe_cfg;

/* env_ids NULL: reset every env.  obs_out [n_env][N][6N] (rows of the reset envs are written). */
int tsm_mpe_spread_reset(const tsm_mpe_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                         const int64_t *env_ids, int64_t n_ids, float *agent_pos, float *agent_vel,
                         float *landmark_pos, int32_t *steps, float *obs_out, void *stream);
/* act i32 [n_env][N] in {0 noop, 1 -x, 2 +x, 3 -y, 4 +y}.  Outputs: obs_next (terminal obs for finished
 * episodes), obs_cur (nullable; what the policy sees next: reset obs where done and auto_reset), rew f32
 * [n_env][N], terminated/truncated u8 [n_env][N], done_env u8 [n_env].  rng_tick (nullable, device
 * u64[1]) += rng_tick_inc once per call. */
int tsm_mpe_spread_step(const tsm_mpe_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                        const int32_t *act, float *agent_pos, float *agent_vel, float *landmark_pos,
                        int32_t *steps, float *obs_next_out, float *obs_cur_out, float *rew_out,
                        uint8_t *terminated_out, uint8_t *truncated_out, uint8_t *done_env_out,
                        int auto_reset, uint64_t *rng_tick, uint64_t rng_tick_inc, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Persistent rollout  [a4 + a7 + a8 fused; SURVEY 8f-1]
 * One launch == one Collector.collect(n_step = n_steps * n_env) on the batched simple_spread env with a
 * shared actor/critic: policy forward + sample + log-prob + value, env step (+ reset of finished envs),
 * buffer index algebra and payload scatter for n_steps vector steps
 * (tianshou/data/collector.py:854-1069 loop body).  Bit-identical to calling tsm_policy_forward ->
 * tsm_mpe_spread_step -> tsm_vrb_add n_steps times.  Additionally stores vnext = V(obs_next) per row
 * (a2c.py:124), so the update needs no critic pass.  All pointers are device pointers; nullable:
 * params (if param_image given), obs_next_store, logp_store, vs_store, vnext_store, obs_cur_out, offset_dev.
 * Per-step outputs: ptr_out/ep_len_out/ep_idx_out i64 [n_steps][n_env], ep_rew_out f64 [n_steps][n_env][N].
 * The sampling counter of step t, row i is  offset + *offset_dev + t*n_env*N + i  (advance it afterwards
 * with tsm_u64_add).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const float *params, *param_image;
    int32_t obs_dim, hidden, n_act, mode; /* mode 1 sample, 2 dist.mode */
    uint64_t policy_seed, offset;
    const uint64_t *offset_dev;
    tsm_mpe_cfg env;
    uint64_t env_seed;
    uint64_t *episode_ctr;
    float *agent_pos, *agent_vel, *landmark_pos;
    int32_t *steps;
    int32_t auto_reset, n_steps;
    float *obs_cur_out;
    void *vrb_state;
    int64_t sub_size;
    uint8_t *done_store;
    float *obs_store, *obs_next_store, *rew_store, *logp_store, *vs_store, *vnext_store;
    int32_t *act_store;
    uint8_t *term_store, *trunc_store;
    int64_t *ptr_out;
    double *ep_rew_out;
    int64_t *ep_len_out, *ep_idx_out;
    /* compact record of the episodes finished during the rollout (nullable; what CollectStats needs, collector.py:
     * 1001-1009, without shipping the dense per-step arrays to the host): i64 words
     *   [n_env] count | [n_env][max_ep] (step << 32 | length) | [n_env][max_ep][N] f64 episode return.
     * count may exceed max_ep (records beyond max_ep are dropped): size max_ep for the worst case. */
    int64_t *ep_rec;
    int32_t max_ep, _pad2;
    /* optional: advance the device-resident sampling counter *offset_dev by offset_inc once every workgroup has read
     * it (done by the last workgroup to finish, counted in *done_ctr: one zero-initialised u32 in HBM that the
     * kernel leaves at zero).  Saves the separate tsm_u64_add launch per collect().  done_ctr NULL = no advance. */
    uint64_t offset_inc;
    uint32_t *done_ctr;
} tsm_rollout_desc;

int tsm_rollout_spread(const tsm_rollout_desc *desc_host, void *stream);
/* The same collect loop for a 128-wide ACTOR (obs -> 128 -> 128 -> 5, e.g. BASELINE configs[2]: N = 8, actor
 * 48-128-128-5 next to a centralized critic that does not fit a CU's LDS): desc.params = the actor's parameters
 * (w0 b0 w1 b1 w2 b2), desc.hidden = 128, param_image unused; vs_store / vnext_store are NOT written -- the update
 * computes V(obs) / V(obs_next) for all rows in two batched passes (a2c.py:121-127).  Everything else as
 * tsm_rollout_spread; results are bit-identical to tsm_mlp_forward -> tsm_categorical_sample -> tsm_mpe_spread_step ->
 * tsm_vrb_add step by step.  Up to eight agents (obs_dim <= 48; TSM_ERR_INVALID above: hosts keep the three-launch loop). */
int tsm_rollout_spread_actor(const tsm_rollout_desc *desc, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batched simple_tag (predator-prey, two teams)  [(f)1, BASELINE configs[4]]
 * Same role and call shape as tsm_mpe_spread_*: the vector-env step the reference performs per env through
 * EnhancedPettingZooEnv (enhanced_pettingzoo_env.py:175-222) under DummyVectorEnv (venvs.py:237-322), for the
 * grouped-policy / self-play / league configurations (training_coordinator.py:413-747).  Agents are ordered
 * adversaries first (n_adv), then good agents; observations are zero-padded to the common width
 * tsm_mpe_tag_obs_dim() = 4 + 2 n_obst + 2 (n_adv + n_good - 1) + 2 n_good (pettingzoo_env.py:55-67 requires
 * identical spaces).  Dynamics restated from the published MPE spec: parity with pettingzoo unpinned.
 * State: agent_pos/vel [n_env][NA][2], landmark_pos [n_env][n_obst][2].
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_env, n_adv, n_good, n_obst, max_cycles, _pad;
    double dt, damping, contact_force, contact_margin;          /* 0.1, 0.25, 100, 1e-3 */
    double adv_size, good_size, obst_size;                       /* 0.075, 0.05, 0.2 */
    double adv_accel, good_accel, adv_speed, good_speed;        /* 3.0, 4.0, 1.0, 1.3 */
} tsm_mpe_tag_cfg;
int tsm_mpe_tag_obs_dim(const tsm_mpe_tag_cfg *cfg_host);
int tsm_mpe_tag_reset(const tsm_mpe_tag_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr,
                      const int64_t *env_ids, int64_t n_ids, float *agent_pos, float *agent_vel,
                      float *landmark_pos, int32_t *steps, float *obs_out, void *stream);
int tsm_mpe_tag_step(const tsm_mpe_tag_cfg *cfg_host, uint64_t seed, uint64_t *episode_ctr, const int32_t *act,
                     float *agent_pos, float *agent_vel, float *landmark_pos, int32_t *steps, float *obs_next_out,
                     float *obs_cur_out, float *rew_out, uint8_t *terminated_out, uint8_t *truncated_out,
                     uint8_t *done_env_out, int auto_reset, uint64_t *rng_tick, uint64_t rng_tick_inc, void *stream);
int tsm_u64_add(uint64_t *counter, uint64_t inc, void *stream);

/* Persistent rollout for simple_tag under GROUPED policies  [a4 + a6 + a7 + a8 fused for BASELINE configs[4]]
 * One launch == one Collector.collect(n_step = n_steps * n_env) on the batched simple_tag env with ONE 64-wide
 * actor/critic per team (FlexibleMultiAgentPolicyManager(mode="grouped"), flexible_policy.py:96-98; collector loop
 * data/collector.py:854-1069): params[0] serves the adversaries (agent columns [0, n_adv)), params[1] the good agents
 * (both flat vectors as for tsm_policy_forward; pass the same pointer twice for one shared policy).  Bit-identical to,
 * per step, tsm_policy_forward on each team's agent-major rows (sampling counter of step t, column a, env e:
 * offset[team] + *offset_dev + t n_env NA + a n_env + e, as MultiAgentPolicy hands a team's columns to its policy)
 * -> tsm_mpe_tag_step -> tsm_vrb_add.  logp_store / vs_store receive each row's own policy outputs (nullable);
 * vnext_store (nullable) V(obs_next) by the row's own team's critic, as tsm_rollout_spread produces it: the next step's V(obs)
 * where the episode goes on, a forward pass of the terminal observation where it ended, a last pass behind the loop for the rows
 * of the final step -- bit-identical to tsm_policy_forward on the stored obs_next rows.  Everything else as tsm_rollout_spread. */
typedef struct {
    const float *params[2];
    uint64_t policy_seed[2], offset[2];
    int32_t mode[2]; /* per team: 1 sample, 2 dist.mode */
    int32_t obs_dim, hidden, n_act;
    int32_t env_major_counter; /* 0: counter a n_env + e inside a step (grouped policies); 1: e NA + a (ONE shared policy
                                * called on the [n_env][NA] rows, params[0] == params[1]) */
    const uint64_t *offset_dev;
    tsm_mpe_tag_cfg env;
    uint64_t env_seed;
    uint64_t *episode_ctr;
    float *agent_pos, *agent_vel, *landmark_pos;
    int32_t *steps;
    int32_t auto_reset, n_steps;
    float *obs_cur_out;
    void *vrb_state;
    int64_t sub_size;
    uint8_t *done_store;
    float *obs_store, *obs_next_store, *rew_store, *logp_store, *vs_store;
    int32_t *act_store;
    uint8_t *term_store, *trunc_store;
    int64_t *ptr_out;
    double *ep_rew_out;
    int64_t *ep_len_out, *ep_idx_out;
    int64_t *ep_rec; /* compact episode record, layout as in tsm_rollout_desc (nullable) */
    int32_t max_ep, _pad2;
    uint64_t offset_inc; /* added to *offset_dev by the last workgroup to finish (done_ctr: zeroed u32 in HBM; NULL = off) */
    uint32_t *done_ctr;
    float *vnext_store; /* [S][n_env][NA] V(obs_next) of every added row (nullable) */
} tsm_rollout_tag_desc;
int tsm_rollout_tag(const tsm_rollout_tag_desc *desc_host, void *stream);

/* ---------------------------------------------------------------------------------------------
 * 128-wide actor, one gradient step in one launch  [a7, a14 at BASELINE configs[2]: actor obs-128-128-A]
 * Replaces the actor half of PPO._update_with_batch (ppo.py:182-212): policy(minibatch).dist, advantage
 * normalisation, ratio / clip / dual-clip surrogate, entropy and the actor's backward pass.  The loss is separable
 * (clip + entropy: actor; value term: critic), so the critic runs beside it: tsm_mlp_forward ->
 * tsm_ppo_loss_fwd_bwd(loss_kind = 2) -> tsm_mlp_backward, and the gradient halves meet in tsm_adam_step.
 * actor_params: w0[H][D] b0[H] w1[H][H] b1[H] w2[A][H] b2[A] (torch parameters() order), H == 128, D <= 64, A <= 16.
 * Sample i of the minibatch is row perm[i] (NULL: first_row + i) of obs [n][D], act, logp_old, adv.
 * Two kernels behind one entry point, picked by M alone (tsm_ppo_actor_rows_grid follows the same rule): 64-sample tiles with the
 * layer-2 weights in registers (csrc/actor_rows64.hip) once every CU gets at least one such tile, 32-sample tiles with all
 * weights in LDS (csrc/ppo_rows.hip) below that; option "actor_tile" = 32 | 64 forces one (tsm_kernel_option_set).
 * n_blocks = tsm_ppo_actor_rows_grid(M) persistent workgroups, each writes ONE gradient slab:
 * grad_slabs_out [n_blocks][tsm_ppo_actor_rows_param_count]; loss_partial_out f64 [n_blocks][4] =
 * {sum clip objective, 0, sum entropy, 0} (the layout tsm_ppo_finalize_many folds).
 * opt_step_dev (nullable, device i64[1]): advanced by one per call, as tsm_ppo_update_fused does.
 * cfg->loss_kind == 1 (policy gradient, obj = logp * adv): logp_old may be NULL, and adv NULL means adv = 1 for every
 * sample -- the launch then yields d(-mean logp)/d params and sum logp (CTDEPolicy.learn's actor term up to the scalar
 * mean(advantage), quirk Q7).
 * ------------------------------------------------------------------------------------------- */
int tsm_ppo_actor_rows_supported(int32_t obs_dim, int32_t hidden, int32_t n_act);
/* Raises the dynamic-LDS limit of every instantiation of the actor / critic rows kernels (idempotent, no stream work).  The
 * launch entry points call it themselves; a host that CAPTURES them into a hipGraph calls it once before its first capture. */
int tsm_ppo_rows_init(void);
int64_t tsm_ppo_actor_rows_param_count(int32_t obs_dim, int32_t hidden, int32_t n_act);
int tsm_ppo_actor_rows_grid(int64_t M);
int tsm_ppo_actor_rows_update(const float *actor_params, int32_t obs_dim, int32_t hidden, int32_t n_act,
                              const float *obs, const int32_t *act, const float *logp_old, const float *adv,
                              const int64_t *perm, int64_t first_row, int64_t M, const float *adv_stats,
                              const tsm_ppo_cfg *cfg, int32_t n_blocks, float *grad_slabs_out,
                              double *loss_partial_out, int64_t *opt_step_dev, void *stream);

/* The critic of the same configuration in one launch: value = MLP(row) with in_dim = n_agent * obs_dim inputs (centralized
 * critic over the joint row, ctde.py:291-294 "concatenate"; n_agent = 1: a local critic on the sample's own observation),
 * the value term of the PPO loss (ppo.py:198-208) for the n_agent samples row * n_agent + a of every row, and the critic's
 * backward pass.  critic_params: w0[H][in_dim] b0[H] w1[H][H] b1[H] w2[1][H] b2[1], H == 128, in_dim <= 384.
 * Row i of the minibatch is joint row rows[i] (NULL: first_row + i) of obs_rows [n][in_dim]; returns / v_s_old are
 * per-sample arrays.  grad_slabs_out [n_blocks][tsm_ppo_critic_rows_param_count]; loss_partial_out f64 [n_blocks][4] =
 * {0, sum of the value-loss terms, 0, 0}; the mean is over Mr * n_agent samples. */
int tsm_ppo_critic_rows_supported(int32_t in_dim, int32_t hidden, int32_t n_agent);
int64_t tsm_ppo_critic_rows_param_count(int32_t in_dim, int32_t hidden);
int tsm_ppo_critic_rows_grid(int64_t Mr);
int tsm_ppo_critic_rows_update(const float *critic_params, int32_t in_dim, int32_t hidden, int32_t n_agent,
                               const float *obs_rows, const float *returns, const float *v_s_old, const int64_t *rows,
                               int64_t first_row, int64_t Mr, const tsm_ppo_cfg *cfg, int32_t n_blocks,
                               float *grad_slabs_out, double *loss_partial_out, void *stream);

/* V(row) of the same critic for MANY rows in one launch, activations never leaving the CU  [a7, a11]
 * Replaces  the critic passes of the on-policy preprocessing, `critic(batch.obs)` / `critic(batch.obs_next)`
 *           (tianshou/algorithm/modelfree/a2c.py:121-127, chunked by max_batchsize there), three GEMM launches here before.
 * critic_params: w0[H][in_dim] b0[H] w1[H][H] b1[H] w2[n_out][H] b2[n_out] (H == 128, in_dim <= 384, n_out <= 16); the value
 * of a row is the MEAN of its n_out outputs (CentralizedCritic: one output per agent, averaged by CTDEPolicy.learn,
 * ctde.py:154-157; n_out = 1: the PPO critic).  Row i is rows[i] (NULL: first_row + i) of obs_rows [n][in_dim]; values_out [Mr].  run_if (nullable, device i32[1]): the launch is a no-op when it holds 0 (a pass a
 * captured graph carries for the rows that need it, tsm_value_next_select).  The first layer sums k in a fixed permuted
 * order (csrc/critic_rows.hip): values agree with tsm_mlp_forward to f32 rounding, not bit for bit.
 * tsm_critic_rows_init: one-time function attributes of the instantiation serving in_dim -- call outside stream capture. */
int tsm_critic_rows_forward_supported(int32_t in_dim, int32_t hidden);
int tsm_critic_rows_init(int32_t in_dim, int32_t hidden);
int tsm_critic_rows_forward(const float *critic_params, int32_t in_dim, int32_t hidden, int32_t n_out,
                            const float *obs_rows, const int64_t *rows, int64_t first_row, int64_t Mr,
                            const int32_t *run_if, float *values_out, void *stream);

/* One gradient step of the same critic in two launches (second generation of tsm_ppo_critic_rows_update)  [a7, a14, a16]
 * (A) tsm_critic_rows_grad_ppo / _td: forward, loss and backward down to dH1 = d loss / d (layer-1 pre-activation), with
 *     the layer-1 weights resident in registers and the observation tile in LDS (csrc/critic_train.hip).
 *       dh1_out [Mr][128] (minibatch row order); rest_slabs_out [n_blocks][P - 128 in_dim]: the gradients of
 *       b1 | W2 | b2 | W3 | b3 (parameter order), one slab per workgroup; n_blocks = tsm_critic_rows_grad_grid(Mr, td).
 *       _ppo with h1_out / dh2_out (nullable pair, [Mr][128] each): the launch PUBLISHES the layer-1 activations and
 *       d loss / d (layer-2 pre-activation) instead of forming dW2 itself -- the W2 part of every rest slab is then NOT written
 *       (hand the optimizer b1 and b2 | W3 | b3 as views into the slabs, and dW2 from launch (B)).
 *     _ppo  Replaces  the value term of PPO._update_with_batch (ppo.py:198-212) on joint rows, as tsm_ppo_critic_rows_update:
 *           loss_partial_out f64 [n_blocks][4] = {0, sum of the value-loss terms, 0, 0}.
 *     _td   Replaces  values = critic(global_obs).mean(1); values_next = critic(global_obs_next).mean(1);
 *           td_target = rew + discount_factor * values_next * (~terminated); critic_loss = mse_loss(values, td_target);
 *           critic_loss.backward()                   (tianshou/algorithm/multiagent/ctde.py:149-172, 188-190)
 *           for CHAINED rows: the batch is the env-major view [E][T] of a time-major store joint_rows [T][E][in_dim] (store
 *           row t * E + e) and obs_next of (e, t) is obs of (e, t + 1) for t < T - 1 -- its value is the next row's, out of
 *           the same forward pass (a tile owns 31 rows and computes the 32nd as a halo); v_last [E] = values of obs_next of
 *           the last slot (tsm_critic_rows_forward on those E rows).  use_full (nullable device i32[1]) != 0: an episode
 *           ended before the last slot, so rows are not chained there -- every target then takes v_next_full[i] (env-major
 *           [T E]: V(obs_next) of every row, a pass the caller launches under the same flag).  rew / terminated of store row r:
 *           x[r * scalar_stride + scalar_offset] (one agent's column of [T][E][N] arrays).  loss_partial_out =
 *           {sum (td_target - values), sum (values - td_target)^2, 0, 0}: mean advantage / critic_loss after division by T E.
 * (B) tsm_critic_rows_dw1: dW1 = dH1^T X over the same rows as a split-K pass (csrc/critic_dw1.hip):
 *       w1_slabs_out [n_chunks][128 in_dim], n_chunks = tsm_critic_rows_dw1_chunks(Mr, in_dim) partial sums over row chunks.
 *     Row i of the minibatch is rows[i], else (tm_T > 0) store row (i % tm_T) * tm_E + i / tm_T, else first_row + i.
 * The optimizer takes both slab arrays as segments (tsm_adam_step_segs: W1 at offset 0, the rest at offset 128 in_dim).
 * critic_params: w0[H][in_dim] b0 w1[H][H] b1 w2[n_out][H] b2[n_out], H == 128, in_dim <= 384 (a multiple of 4 above 64).
 * w1_image (nullable): w0 in the kernels' FRAGMENT ORDER, [8 waves][kj][64 lanes][4] floats with element (w, j, lane, i) =
 * w0[16 w + lane % 16][16 j + 4 (lane / 16) + i] (zero where the column is >= in_dim), kj = tsm_critic_rows_w1_image_kj(in_dim):
 * every wave then loads its share with 1 KB-coalesced instructions instead of gathering 64-B pieces of sixteen 1.5 KB rows
 * (3.8 us of a 10 us prologue at in_dim = 384).  tsm_critic_rows_w1_image builds it from w0; tsm_adam_step_segs keeps it in
 * step with the parameters (tsm_slab_seg.frag_image).  It must hold the SAME values as critic_params' w0; NULL = gather. */
int64_t tsm_critic_rows_param_count(int32_t in_dim, int32_t hidden, int32_t n_out);
int tsm_critic_rows_w1_image_kj(int32_t in_dim);
int64_t tsm_critic_rows_w1_image_elems(int32_t in_dim);
int tsm_critic_rows_w1_image(const float *w0, int32_t in_dim, float *image_out, void *stream);
int tsm_critic_rows_grad_grid(int64_t Mr, int32_t td);
int tsm_critic_rows_grad_ppo(const float *critic_params, const float *w1_image, int32_t in_dim, int32_t hidden, int32_t n_agent,
                             const float *obs_rows, const float *returns, const float *v_s_old, const int64_t *rows,
                             int64_t first_row, int64_t Mr, const tsm_ppo_cfg *cfg, int32_t n_blocks, float *dh1_out,
                             float *h1_out, float *dh2_out, float *rest_slabs_out, double *loss_partial_out, void *stream);
int tsm_critic_rows_grad_td(const float *critic_params, const float *w1_image, int32_t in_dim, int32_t hidden, int32_t n_out,
                            const float *joint_rows, int64_t T, int64_t E, const float *rew, const uint8_t *terminated,
                            int64_t scalar_stride, int64_t scalar_offset, const float *v_last, const float *v_next_full,
                            const int32_t *use_full, double gamma, int32_t n_blocks, float *dh1_out,
                            float *rest_slabs_out, double *loss_partial_out, void *stream);
/* The two scalars of CTDEPolicy.learn and the actor gradient's scale from the kernels' partial sums (ctde.py:181-199):
 * critic_partial [nb_c][4] = {sum (td - v), sum (v - td)^2, ..} (tsm_critic_rows_grad_td), actor_partial [nb_a][4] =
 * {sum log_probs, ..} (tsm_ppo_actor_rows_update, loss_kind 1, adv NULL) -> scalars_out[2] = {actor_loss =
 * -mean(log_probs) * mean(advantage), critic_loss}, mean_adv_out[1] = mean(advantage) (the scale_dev of the actor's slab
 * segment).  One workgroup; fixed summation order.  scalars_out may be pinned host memory. */
int tsm_ctde_finalize(const double *critic_partial, int32_t nb_c, const double *actor_partial, int32_t nb_a, int64_t B,
                      float *scalars_out, float *mean_adv_out, void *stream);
int tsm_critic_rows_dw1_chunks(int64_t Mr, int32_t in_dim);
/* side (nullable, at most 2): gradient-slab sets of OTHER parameter segments, already complete when this launch starts (the
 * actor step's slabs, the small-gradient slabs of launch (A)), summed to ONE row each by extra workgroups of this launch while
 * its main workgroups wait on their loads: out[i] = sum over slabs of slabs[s * stride + i] in tsm_adam_step's own order, so
 * handing `out` to the optimizer as a one-slab segment gives bit-identical gradients -- and the optimizer launch reads 6 MB
 * instead of 48 (BASELINE configs[2] step). */
typedef struct tsm_slab_reduce {
    const float *slabs;
    int64_t n, stride;
    int32_t n_slab, _pad;
    float *out;
} tsm_slab_reduce;
/* dh2 / h1 / w2_slabs_out (nullable triple): the SECOND layer's weight gradient as extra column blocks of the same launch,
 * dW2 = dH2^T H1 over the same row chunks, from the activations launch (A) published in minibatch order
 * (tsm_critic_rows_grad_ppo: h1_out, dh2_out): w2_slabs_out [n_chunks][128 x 128].  Launch (A) then keeps no W2 gradient --
 * a rank-32 update per 32-row tile stored as a 64 KB slab (17 MB per BASELINE configs[2] step) becomes 32 chunk slabs (2 MB). */
int tsm_critic_rows_dw1(const float *dh1, const float *obs_rows, int32_t in_dim, const int64_t *rows, int64_t first_row,
                        int64_t tm_T, int64_t tm_E, int64_t Mr, int32_t n_chunks, float *w1_slabs_out,
                        const float *dh2, const float *h1, float *w2_slabs_out,
                        const tsm_slab_reduce *side, int32_t n_side, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Latency-grade all-reduce of a small vector over peer-mapped memory  [SURVEY 8e: the 45 KB shared-policy gradient]
 * Replaces  torch.distributed.all_reduce(flat_grad) in front of Optimizer.step (algorithm_base.py:485-498) for env-sharded
 *           replicas on one node (the reference itself has no distributed path: utils/net/common.py:477-519).
 * One-shot write-to-peers over IPC-mapped fine-grained memory (every element one 8-byte store {value, call stamp}: no fences),
 * one launch per call, rank-ordered sum (bit-identical on every rank); csrc/p2p.hip has the protocol.  Setup: every rank tsm_p2p_create -> tsm_p2p_export -> the ranks exchange
 * the tsm_p2p_ipc_handle_bytes()-byte handles by any channel (the host binding uses the process group) -> tsm_p2p_import of
 * every peer -> tsm_p2p_handshake on every rank (one stamped word per peer each way, bounded spin, outside any capture); the
 * ranks then AGREE (over their process group) to use this path only if every rank's handshake passed -- otherwise each destroys
 * its handle and the backend's own all-reduce serves for the rest of the process (the host binding: TSM_P2P_ALLREDUCE unset =
 * "the handshake decides", 1 = required, 0 = off).
 * Failure behaviour (fail-stop): a peer whose word does not arrive within the spin limit (tsm_p2p_set_timeout; ~2 s) sets the
 * handle's error word; the element that timed out is NOT updated (tsm_p2p_adam_step leaves parameter and moments as they
 * were, tsm_p2p_all_reduce leaves the element), every later launch on the handle is a no-op, and tsm_p2p_failed (synchronises
 * the device) returns 1: the host raises.  No hang, and no replica ever steps on a partial sum.
 * ------------------------------------------------------------------------------------------- */
int64_t tsm_p2p_ipc_handle_bytes(void);
int tsm_p2p_create(int32_t rank, int32_t world, int64_t max_floats, void **handle_out);
int tsm_p2p_export(void *handle, void *ipc_handle_out);
int tsm_p2p_import(void *handle, int32_t peer, const void *ipc_handle);
int tsm_p2p_all_reduce(void *handle, float *data, int64_t n, void *stream);
/* tsm_reduce_slabs (scale 1 / world) + tsm_p2p_all_reduce + tsm_adam_step (without a gradient-norm clip) in ONE launch: every
 * workgroup sums the slabs of its own slice of the parameter vector, runs the one-shot protocol on that slice and applies Adam to
 * it (the three steps have no grid-wide dependency), so the replicas' gradient step is one launch like the single-GPU step.
 * Bit-identical to the three-launch form.  Arguments as tsm_adam_step; n <= the handle's max_floats. */
int tsm_p2p_adam_step(void *handle, float *param, const float *grad_slabs, int32_t n_slab, int64_t n, float *exp_avg,
                      float *exp_avg_sq, int64_t step, const int64_t *step_dev, double lr, const double *lr_dev, double beta1,
                      double beta2, double eps, double weight_decay, float *param_image, const int32_t *image_map, void *stream);
int tsm_p2p_set_timeout(void *handle, double seconds);
int tsm_p2p_handshake(void *handle, int32_t *ok_out, void *stream);
int tsm_p2p_failed(void *handle);
/* The error word copied asynchronously into PINNED host memory behind the work queued on `stream`: valid once the host has waited
 * for that point of the stream.  The host binding queues it in front of the event of every update's loss statistics and raises
 * on every rank when it reads them (no extra synchronisation; algorithm/ppo.py, ppo_generic.py). */
int tsm_p2p_error_async(void *handle, int32_t *host_pinned_out, void *stream);
int tsm_p2p_destroy(void *handle);

/* ---------------------------------------------------------------------------------------------
 * CTDE global state  [a16]
 * Replaces  GlobalStateConstructor.build("concatenate" | "mean")
 *           (tianshou/algorithm/multiagent/ctde.py:291-300) for per-agent arrays [B][D] given in
 *           env.agents order (quirk Q5).  mode 0: out [B][N*D] concat; mode 1: out [B][D] mean.
 * With the joint-lane layout obs[..][N][D] the concatenation is a free reshape; this entry point
 * serves hosts that hold one array per agent (the reference's dict-of-tensors AoS).
 * ------------------------------------------------------------------------------------------- */
int tsm_global_state(const float *const *obs_by_agent_host, int32_t n_agent, int64_t B, int32_t D,
                     int mode, float *out, void *stream);

/* Minibatch permutations  [a14]
 * Replaces  the np.random.permutation(length) drawn by Batch.split(shuffle=True) once per repeat
 *           (tianshou/data/batch.py:1219 via ppo.py:179) when shuffling is done on the device.
 * Writes n_perm pseudo-random permutations of [0, n) in ONE launch (keyed Feistel network + cycle walking, key =
 * Philox(seed, counter + *counter_dev + p)):  out[p][i] = pi_p(i) * scale + (p / group_size) * offset_mul.
 * counter_dev (nullable, device u64[1]) lets a captured hipGraph draw fresh permutations on every replay
 * (advance it with tsm_u64_add). */
int tsm_random_permutations(int64_t n, int32_t n_perm, uint64_t seed, uint64_t counter, const uint64_t *counter_dev,
                            int64_t scale, int32_t group_size, int64_t offset_mul, int64_t *out, void *stream);
/* The same draw keyed by *counter_dev alone, after which the LAST workgroup to finish adds counter_inc to *counter_dev (every
 * workgroup has read it by then): a captured graph needs no tsm_u64_add launch behind its draws.  done_ctr: a zeroed u32 in HBM
 * owned by this call site (left at zero again). */
int tsm_random_permutations_advance(int64_t n, int32_t n_perm, uint64_t seed, uint64_t *counter_dev, uint64_t counter_inc,
                                    uint32_t *done_ctr, int64_t scale, int32_t group_size, int64_t offset_mul, int64_t *out,
                                    void *stream);

/* CTDEPolicy.learn loss head  [a16]
 * Replaces  the TD target / critic MSE / policy-gradient arithmetic of CTDEPolicy.learn
 *           (tianshou/algorithm/multiagent/ctde.py:149-185) between the network forwards and `backward()`.
 * q, q_next [B][n_out]: centralized critic on global_obs / global_obs_next (mean over n_out = the value, :154-157);
 * logits [B][n_act]: decentralized actor on the local obs; act i64 [B].  The (B,) x (B,1) broadcast of :185 is kept
 * (actor_loss = -mean(log_probs) * mean(advantage), quirk Q7).  Outputs: dq [B][n_out] = d critic_loss / d q,
 * dlogits [B][n_act] = d actor_loss / d logits, scalars[2] = {actor_loss, critic_loss}.
 * partial: tsm_ctde_head_partial_elems(B) doubles of scratch. */
int64_t tsm_ctde_head_partial_elems(int64_t B);
int tsm_ctde_td_head(const float *q, const float *q_next, int32_t n_out, const float *rew,
                     const uint8_t *terminated, float gamma, const float *logits, const int64_t *act,
                     int32_t n_act, int64_t B, float *dq, float *dlogits, double *partial, float *scalars,
                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fully-connected networks of arbitrary width (f32 MFMA tiled GEMMs)  [a7, a16]
 * Replaces  nn.Linear / activation stacks and autograd through them for actor / critic modules that do not fit
 *           the fused 64-wide kernels: DecentralizedActor.forward (ctde.py:366-379), CentralizedCritic.forward
 *           (ctde.py:402-414), MLP (tianshou/utils/net/common.py:67-160), and `loss.backward()` (ctde.py:188-194).
 * params: flat f32 vector in torch `parameters()` order  w0[dims[1]][dims[0]], b0[dims[1]], w1, b1, ...
 * `act` (1 relu, 2 tanh, 0 none) follows every layer except the last.
 * tsm_mlp_forward : acts = consecutive blocks [B][dims[1]], [B][dims[2]], ... (the last block is the output);
 *                   tsm_mlp_act_elems(desc, B) floats.
 * tsm_mlp_backward: d_out [B][dims[L]] = gradient w.r.t. the output; writes n_split gradient slabs
 *                   slabs[n_split][n_param] (batch split; sum them with tsm_adam_step / tsm_reduce_slabs);
 *                   d_acts: workspace of tsm_mlp_act_elems floats (hidden-layer gradients).
 * ------------------------------------------------------------------------------------------- */
#define TSM_MLP_MAX_LAYERS 8
typedef struct tsm_mlp_desc {
    int32_t n_layers;
    int32_t act;
    int32_t dims[TSM_MLP_MAX_LAYERS + 1];
} tsm_mlp_desc;
int64_t tsm_mlp_param_count(const tsm_mlp_desc *desc);
int64_t tsm_mlp_act_elems(const tsm_mlp_desc *desc, int64_t B);
int tsm_mlp_forward(const tsm_mlp_desc *desc, const float *params, const float *x, int64_t B, float *acts,
                    void *stream);
/* tsm_mlp_forward whose launches are no-ops when *run_if_nonzero == 0 (device i32; read by every workgroup): lets a
 * captured graph hold a pass that is only needed for some inputs (see tsm_value_next_select). */
int tsm_mlp_forward_cond(const tsm_mlp_desc *desc, const float *params, const float *x, int64_t B, float *acts,
                         const int32_t *run_if_nonzero, void *stream);
int tsm_mlp_backward(const tsm_mlp_desc *desc, const float *params, const float *x, int64_t B, const float *acts,
                     const float *d_out, float *d_acts, int32_t n_split, float *slabs, int64_t slab_stride,
                     void *stream);
/* slab_stride: distance in floats between consecutive slabs (0 = this net's parameter count).  A larger stride lets
 * several networks write their gradients side by side into the slabs of ONE joint parameter vector (actor + critic
 * under one optimizer with a global gradient-norm clip, algorithm_base.py:485-498). */

#ifdef __cplusplus
}
#endif
#endif /* TSMARL_H */
