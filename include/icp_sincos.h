/*
 * icp_sincos.h — ONE sine/cosine for the three places that turn the Euler angles of ModelFittingParameters into a rotation matrix
 * (api/sampling/ModelFittingParameters.scala:79-86: Rotation(phi, theta, psi, centre)): the library's host side
 * (icp-proposal_amd/csrc/icp_abi.hip: pose_from_theta), its device code (the pose walks of the on-device Metropolis–Hastings loop,
 * api/sampling/proposals/PoseProposals.scala:31-62, build their proposed pose on the GPU) and the CPU oracle
 * (oracle/icp_oracle.c: orc_rotation_matrix).
 *
 * Why not libm: the device's sin/cos are a different implementation from the host's, and a rotation matrix that differs in its last
 * bit moves every model point — the correspondence indices are compared BIT FOR BIT between the three.  This function is plain IEEE
 * double arithmetic — additions, multiplications, one rounding each, in a fixed order, no fused multiply-add (every translation unit
 * that includes it is compiled with -ffp-contract=off), no table, no library call — so the same source gives the same bits wherever
 * it is compiled.
 *
 * Algorithm (the classic one; constants as published with fdlibm's k_sin.c / k_cos.c / e_rem_pio2.c): n = nearest integer to
 * x·2/π, three-part Cody–Waite reduction y = ((x − n·P1) − n·P2) − n·P3 to |y| <= π/4 (+ a little), minimax polynomials for
 * sin y and cos y on that interval, quadrant from n mod 4.  Absolute error <= 2.3e-16 against libm for |x| up to 1e4 (checked over
 * 500,000 arguments; pose angles are fractions of a radian); far beyond that the reduction loses bits gracefully — deterministic
 * everywhere, which is what matters here.  Non-finite or absurdly large x (|x| > 6e15) -> NaN.
 */
#ifndef ICP_SINCOS_H
#define ICP_SINCOS_H

#if defined(__HIPCC__)
#define ICP_SINCOS_FN __host__ __device__ static inline
#else
#define ICP_SINCOS_FN static inline
#endif

ICP_SINCOS_FN void icp_sincos(double x, double *s_out, double *c_out) {
  /* π/2 in three parts: P1 has 33 significant bits (n·P1 exact for |n| < 2^20), P2 the next 33, P3 the rest */
  const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624871116645580e-21;
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  if (!(x - x == 0.0)) { *s_out = x - x; *c_out = x - x; return; } /* inf, NaN -> NaN */
  double t = x * TWO_OVER_PI;
  if (!(t < 4.0e15 && t > -4.0e15)) { *s_out = (x - x) / (x - x); *c_out = *s_out; return; } /* beyond any angle: NaN, everywhere */
  /* nearest integer, halves away from zero: (long long) truncates toward zero */
  long long n = (long long)(t >= 0.0 ? t + 0.5 : t - 0.5);
  double fn = (double)n;
  double y = x - fn * P1;
  y = y - fn * P2;
  y = y - fn * P3;
  {
    const double z = y * y;
    /* sin y = y + y·z·(S1 + z·(S2 + z·(S3 + z·(S4 + z·(S5 + z·S6))))) */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    /* cos y = 1 − z/2 + z·z·(C1 + z·(C2 + z·(C3 + z·(C4 + z·(C5 + z·C6))))) */
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S5 + z * S6;
    ps = S4 + z * ps;
    ps = S3 + z * ps;
    ps = S2 + z * ps;
    ps = S1 + z * ps;
    const double sy = y + (y * z) * ps;
    double pc = C5 + z * C6;
    pc = C4 + z * pc;
    pc = C3 + z * pc;
    pc = C2 + z * pc;
    pc = C1 + z * pc;
    const double cy = (1.0 - 0.5 * z) + (z * z) * pc;
    switch ((int)(n & 3)) {
      case 0: *s_out = sy; *c_out = cy; break;
      case 1: *s_out = cy; *c_out = -sy; break;
      case 2: *s_out = -sy; *c_out = -cy; break;
      default: *s_out = -cy; *c_out = sy; break;
    }
  }
}

/* Rotation(phi, theta, psi, centre) as the library poses with it: R = Rz(phi)·Ry(theta)·Rx(psi), row-major
 * ([SCALISMO-UNVERIFIED]; a caller with Scalismo's own matrix registers it: icp_ctx_set_rotation) */
ICP_SINCOS_FN void icp_rotation_matrix(double phi, double theta, double psi, double *R) {
  double cph, sph, cth, sth, cps, sps;
  icp_sincos(phi, &sph, &cph);
  icp_sincos(theta, &sth, &cth);
  icp_sincos(psi, &sps, &cps);
  R[0] = cth * cph; R[1] = sps * sth * cph - cps * sph; R[2] = sps * sph + cps * sth * cph;
  R[3] = cth * sph; R[4] = cps * cph + sps * sth * sph; R[5] = cps * sth * sph - sps * cph;
  R[6] = -sth;      R[7] = sps * cth;                   R[8] = cps * cth;
}

#endif /* ICP_SINCOS_H */
