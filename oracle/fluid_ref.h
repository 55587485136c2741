/*
 * fluid_ref.h — kernel-level entry points of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it.  libfluid_hip.so never links or calls it.
 *
 * Each function is a serial restatement of ONE __global__ kernel of
 * /root/reference/FluidSimulation/kernel.cu applied to every cell of an n x n
 * grid (x fastest, index = x + y*n), with the texture semantics of SURVEY.md
 * Appendix B restated as plain loads.  Inputs and outputs never alias except in
 * ref_add_force / ref_add_vorticity (the reference RMWs through a surface).
 *
 * Parity status: the reference has no tests, golden vectors or fixtures
 *  and cannot be built here (CUDA-only, section 8c), so
 * this oracle is pinned only by the hand-derivable known-answer tests of
 * SURVEY.md section 4 (tests/test_oracle_kat.py) => "parity unpinned".
 */
#ifndef FLUID_REF_H
#define FLUID_REF_H

#ifdef __cplusplus
extern "C" {
#endif

#define REF_API __attribute__((visibility("default")))

/* kernel.cu:114-140, input filter LINEAR (kernel.cu:546). vel: float2[n*n]. */
REF_API void ref_advection(int n, float dt, const float *vel_in, float *vel_out);
/* kernel.cu:142-165, velocity POINT (:669), density LINEAR (:672). dens: float4[n*n]. */
REF_API void ref_advection_density(int n, float dt, const float *vel, const float *dens_in,
                                   float *dens_out);
/* kernel.cu:173-205, POINT (:589): ONE sweep. */
REF_API void ref_diffusion(int n, float dt, float viscosity, const float *vel_in, float *vel_out);
/* kernel.cu:207-230 (filter LINEAR :560, fetched at texel centres => exact texels). */
REF_API void ref_vorticity(int n, const float *vel, float *vort);
/* kernel.cu:232-263 (vorticity LINEAR :571 at texel centres; velocity RMW in place). */
REF_API void ref_add_vorticity(int n, float dt, const float *vort, float *vel);
/* kernel.cu:265-289, POINT (:608). */
REF_API void ref_divergence(int n, const float *vel, float *div);
/* kernel.cu:291-324, POINT (:625, :628): ONE sweep. */
REF_API void ref_pressure_jacobi(int n, const float *p_in, const float *div, float *p_out);
/* kernel.cu:326-355, POINT (:646, :649). */
REF_API void ref_projection(int n, const float *vel_in, const float *p, float *vel_out);
/* kernel.cu:357-383: in-place RMW of velocity and density over ALL cells. */
REF_API void ref_add_force(int n, float dt, float radius, float sx, float sy, float fx, float fy,
                           const float rgba[4], float *vel, float *dens);


/* The same rules on an nx x ny grid (index = x + y*nx), extent taken per axis: SURVEY.md 8f-4.
 * ref_X(n, ...) == ref_X_rect(n, n, ...); for nx != ny there is no reference behaviour (every
 * kernel of kernel.cu takes one gridResolution), these define it. */
REF_API void ref_advection_rect(int nx, int ny, float dt, const float *vel_in, float *vel_out);
REF_API void ref_advection_density_rect(int nx, int ny, float dt, const float *vel, const float *dens_in,
                                        float *dens_out);
REF_API void ref_diffusion_rect(int nx, int ny, float dt, float viscosity, const float *vel_in, float *vel_out);
REF_API void ref_vorticity_rect(int nx, int ny, const float *vel, float *vort);
REF_API void ref_add_vorticity_rect(int nx, int ny, float dt, const float *vort, float *vel);
REF_API void ref_divergence_rect(int nx, int ny, const float *vel, float *div);
REF_API void ref_pressure_jacobi_rect(int nx, int ny, const float *p_in, const float *div, float *p_out);
REF_API void ref_projection_rect(int nx, int ny, const float *vel_in, const float *p, float *vel_out);
REF_API void ref_add_force_rect(int nx, int ny, float dt, float radius, float sx, float sy, float fx, float fy,
                                const float rgba[4], float *vel, float *dens);

#ifdef __cplusplus
}
#endif
#endif
