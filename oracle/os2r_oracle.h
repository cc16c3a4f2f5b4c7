/* os2r_oracle.h — CPU restatement of the gym-os2r env-step path (TEST INFRASTRUCTURE ONLY;
 * see the header of os2r_oracle.c).  Host pointers everywhere, fp64 only. */
#ifndef OS2R_ORACLE_H_
#define OS2R_ORACLE_H_
#include <stdint.h>
#include "../include/os2r.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcSim OrcSim;

/* counter RNG */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_uniform2(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr, uint32_t blk, double u[2]);

/* task-level pieces (pinned by tests/golden/) */
double orc_tolerance(double x, double lower, double upper, double margin, int sigmoid, double value_at_margin);
void orc_leg_joint_angles(const double def6[6], double pitch, double out[2]);
double orc_wrap(double x);
void orc_observe(const Os2rTaskSpec* ts, const double* q, const double* qd, const double hist1[2], double* obs);
int orc_done(const Os2rTaskSpec* ts, const double* obs);
double orc_reward(const Os2rTaskSpec* ts, const double* obs, const double a0[2], const double a1[2]);

/* dynamics of one environment (parity unpinned; checked by known answers + invariants) */
void orc_dynamics(const Os2rModel* md, double dt, const double* mass_scale, const double* damping,
                  double gravity_z, const double* q, const double* qd, const double* tau_full,
                  double* qdd, double* minv, double* rw, double* ow);
void orc_contact_points(const Os2rModel* md, double margin, const double (*rw)[9], const double (*ow)[3],
                        int* active, double (*pw)[3], double* gap);
void orc_substep(const Os2rConfig* cfg, const double* mass_scale, const double* damping, const double* friction,
                 const double* mu, double gravity_z, double* q, double* qd, const double tau2[2]);

/* contact models of the oracle: the specification, and an oracle-only comparison model (os2r_oracle.c) */
enum { ORC_CONTACT_CENTROID = 0, ORC_CONTACT_PER_VERTEX = 1 };
void orc_substep_model(const Os2rConfig* cfg, int contact_model, const double* mass_scale, const double* damping,
                       const double* friction, const double* mu, double gravity_z, double* q, double* qd,
                       const double tau2[2]);
int orc_contact_problem(const Os2rConfig* cfg, int contact_model, const double* mass_scale, const double* damping,
                        const double* friction, const double* mu, double gravity_z, const double* q,
                        const double* qd, const double tau2[2], int max_rows, double* vstar, double* minv,
                        double* J, double* target, int32_t* kind, int32_t* normal_row, int32_t* body,
                        double* bound, double* box, double* lambda, double* point, double* v_out);

/* batched simulator mirroring include/os2r.h on host arrays */
int orc_create(const Os2rConfig* cfg, OrcSim** out);
void orc_destroy(OrcSim* s);
void orc_set_threads(OrcSim* s, int n);
void orc_set_contact_model(OrcSim* s, int model);
int orc_get_solver_counts(OrcSim* s, int8_t* sweeps, int8_t* solves);   /* diagnostics: [substeps][N] each, of the last step; the first call switches recording on */
int orc_get_small_solve_counts(OrcSim* s, int8_t* small);               /* of those solves: the dual solves of small free sets */
#ifdef ORC_EXPERIMENTS
/* The laboratory build only (make lab -> liboracle_lab.so; tests/diag/ and the studies under docs/studies/): switches that change
 * what the solver does.  The checker library (libos2r_oracle.so) has none of them: there they are the specification's constants. */
void orc_set_experimental_block_solve(int on);   /* oracle-only experiments, see os2r_oracle.c */
void orc_set_experimental_row_order(int order);
void orc_set_experimental_rounds(int max_rounds, int stop_at_cap);
long long orc_debug_counter(int which, int reset);   /* diagnostics, see os2r_oracle.c */
void orc_debug_free_set_hist(long long* out16x12, int reset);   /* diagnostics: free-set shapes by solve index (the first orc_debug_counter call switches recording on) */
void orc_set_experimental_lag_box(int on);   /* studies: 1 -- the lagged friction box (no phase 1 for an environment that remembers every active row) */
void orc_set_experimental_incons(double threshold);   /* studies: the inconsistent-free-set test's threshold (1e-4) */
void orc_set_experimental_incons_once(int n);   /* studies: at most n inconsistent-set steps per iteration (0: no limit) */
void orc_set_experimental_prox(int k);   /* studies: proximal iterations of the regularised solve (2; 3 up to round 5) */
void orc_set_experimental_solve_first(int k);   /* studies (round 5): k > 0 -- an iteration whose predecessor in the env-step took >= k solves opens with a solve instead of the first sweeps */
void orc_set_experimental_margin(double m);   /* studies (round 5): free rows keep the fraction m of their box width away from the bounds */
void orc_set_experimental_box_probe(int on);  /* studies (round 5): accumulate how far the box-fixing normal impulses are from the converged normal-only solve and from the final ones */
void orc_debug_box_stat(double* out4);
void orc_set_experimental_warm_p0(int on);   /* studies (round 5): 1 -- phase 1 starts from the remembered normal / joint-friction impulses */
void orc_set_experimental_snap(int on);   /* studies (round 5): 1 -- the warm start keeps a tangential row that ended on a bound on the same bound of the new box */
void orc_set_experimental_multicut(int k);   /* studies (round 5): k > 0 -- a step cut below 1e-k of its length pins every row within 10 x that fraction of its bound at once */
void orc_set_experimental_repin(int on);   /* studies (round 5): 1 -- a cut step puts back, at once, every row the last sweep released from a bound and that violates it again */
void orc_set_experimental_equil(int on);   /* studies: 1 (the specification since round 5) -- the regularised solve weighs every free row with 1 / |g_r|^2; 0: round 4 */
void orc_set_experimental_pivot(int on);   /* studies (round 5): 1 -- an inconsistent-set step that would pin a row it pinned before in this iteration ends phase 2 */
void orc_set_experimental_prox_later(int k);   /* studies: the same for an environment's second and later solves of an iteration (0: as the first) */
void orc_set_experimental_stall(double factor);
void orc_set_experimental_sweep_after_cut(int on);
void orc_set_experimental_clamp_all(int on);
void orc_set_experimental_block_kind(int kind);   /* 0: enumeration of the block's active sets, 1: Gauss-Seidel pass + one exact solve of the rows left free */
void orc_set_experimental_warm(int mode, int first);  /* studies: 1 the specification, 0 no warm start, 2 round 3 (forgotten between env-steps); K sweeps before the first check */
void orc_set_experimental_small(int max_rows);   /* studies: the dual solve serves free sets of up to that many rows (0: every solve is the regularised one -- the specification; at most 3) */
#endif
int orc_get_solver_state(OrcSim* s, double* lam, uint32_t* flags);   /* layout of os2r_get_solver_state (include/os2r.h) */
int orc_set_solver_state(OrcSim* s, const double* lam, const uint32_t* flags);
int orc_reset(OrcSim* s, const uint8_t* mask, double* obs);
int orc_step(OrcSim* s, const double* actions, double* obs, double* reward, uint8_t* done, double* term_obs);
int orc_get_state(OrcSim* s, double* q, double* qd);
int orc_set_state(OrcSim* s, const double* q, const double* qd);
int orc_get_action_history(OrcSim* s, int which, double* out);
int orc_set_action_history(OrcSim* s, int which, const double* in);
int orc_set_params(OrcSim* s, int field, const double* src);
int orc_get_params(OrcSim* s, int field, double* dst);
int orc_get_episode_info(OrcSim* s, int32_t* steps, uint32_t* episode, uint8_t* pose);
int orc_set_episode_info(OrcSim* s, const int32_t* steps, const uint32_t* episode, const uint8_t* pose);
uint64_t orc_get_step_count(OrcSim* s);
void orc_set_step_count(OrcSim* s, uint64_t v);

#ifdef __cplusplus
}
#endif
#endif
