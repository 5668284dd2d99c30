/* orc_interpv_tmpl.h -- TEST INFRASTRUCTURE (CPU oracle), body of the 1-D vertical interpolation package.
 *
 * Included twice by orc_interpv.c, as the reference compiles every *_Body.inc twice (`#define real48 single`
 * / `double`, e.g. src/interpv/Interp1D_Linear.F90:36-38, :54-56):
 *     REAL   float | double                (real48)
 *     FN(x)  x | x##8                       (Interp1D_Linear -> orc_interp1d_linear, Interp1D_Linear8 -> ..._linear8)
 *
 * Arrays are the reference's: Fortran order, (ijDim, levels), column index fastest; posnDestInSrc holds
 * 1-based level indices.  All arithmetic is in the type and in the operation order of the reference text,
 * compiled without contraction (-ffp-contract=off, oracle/Makefile).
 */
#define VLS(i, k) vls[(size_t)(i) + (size_t)((k) - 1) * (size_t)sij]      /* vLevelSource(i+1, k)          */
#define SS(i, k)  ss[(size_t)(i) + (size_t)((k) - 1) * (size_t)sij]       /* stateSource(i+1, k)           */
#define SDS(i, k) sds[(size_t)(i) + (size_t)((k) - 1) * (size_t)sij]      /* stateDerivSource(i+1, k)      */
#define AT(a, i, t) a[(size_t)(i) + (size_t)(t) * (size_t)dij]            /* dest arrays (i+1, t+1)        */

/* Interp1D_FindPos_Body.inc:69-144.  indexReal / uncertainty are default REAL whatever real48 is (:69-71);
 * the direction of the levels is read from column 1 only (:88, :123). */
void FN(orc_interp1d_findpos)(int n, int ns, int nd, int sij, int dij, const REAL *vls, int *posn, const REAL *vld)
{
    float *idx = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int t = 0; t < nd; t++) {
        for (int s = 0; s < n; s++) idx[s] = 0.5f * (float)(ns + 1);                    /* :84-86 */
        float unc = 0.5f * (float)(ns - 1);                                              /* :87    */
        float sgn = (VLS(0, 2) > VLS(0, 1)) ? 1.0f : -1.0f;                              /* :90 / :128 */
        while (unc > 0.95f) {                                                            /* :103, :129 */
            unc = 0.5f * unc;
            for (int s = 0; s < n; s++) {
                int ii = (int)idx[s];                                                    /* :115 truncation */
                REAL v = AT(vld, s, t);
                if (v >= VLS(s, ii)) idx[s] = idx[s] + sgn * unc;                        /* :116-117 / :133-134 */
                if (v <= VLS(s, ii + 1)) idx[s] = idx[s] - sgn * unc;                    /* :118-119 / :135-136 */
            }
        }
        for (int s = 0; s < n; s++) AT(posn, s, t) = (int)idx[s];                        /* :142 */
    }
    free(idx);
}

/* Interp1D_NearestNeighbour_Body.inc:68-97 */
int FN(orc_interp1d_nearestneighbour)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                                      const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup)
{
    (void)ns; (void)sds; (void)sdd; (void)xdown; (void)xup; (void)gdown; (void)gup;
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            int p = AT(posn, i, t);
            REAL lb = VLS(i, p), la = VLS(i, p + 1), sa = SS(i, p + 1), sb = SS(i, p), v = AT(vld, i, t);
            REAL db = v - lb, da = v - la;
            if (db < 0) db = -db;
            if (da < 0) da = -da;
            AT(sd, i, t) = (db < da) ? sb : sa;                                          /* :82-87 */
        }
    return 0;
}

/* the clamp that closes Linear, CubicLagrange and CubicWithDerivs (Interp1D_Linear_Body.inc:101-108) */
#define CLAMP_TAIL(i, t, WITH_DERIV)                                                               \
    if (!xdown && AT(vld, i, t) < VLS(i, smallest)) {                                              \
        AT(sd, i, t) = SS(i, smallest); if (WITH_DERIV) AT(sdd, i, t) = SDS(i, smallest);          \
    } else if (!xup && AT(vld, i, t) > VLS(i, largest)) {                                          \
        AT(sd, i, t) = SS(i, largest); if (WITH_DERIV) AT(sdd, i, t) = SDS(i, largest);            \
    }

/* Interp1D_Linear_Body.inc:67-112; returns -1 where the reference logs an error and returns (:67-70) */
int FN(orc_interp1d_linear)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                            const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup)
{
    (void)sds; (void)sdd; (void)gdown; (void)gup;
    if (ns < 2) return -1;
    int smallest = (VLS(0, 1) < VLS(0, 2)) ? 1 : ns, largest = (VLS(0, 1) < VLS(0, 2)) ? ns : 1;      /* :73-81 */
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            int p = AT(posn, i, t);
            REAL lev1 = VLS(i, p), s1 = SS(i, p), lev2 = VLS(i, p + 1), s2 = SS(i, p + 1);
            REAL slope = (s2 - s1) / (lev2 - lev1);                                      /* :97 */
            REAL prod = slope * (AT(vld, i, t) - lev1);
            AT(sd, i, t) = prod + s1;                                                    /* :98 */
            CLAMP_TAIL(i, t, 0)
        }
    return 0;
}

/* Interp1D_CubicLagrange_Body.inc:88-157: REAL(double) arithmetic whatever real48 is (:70-74) */
int FN(orc_interp1d_cubiclagrange)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                                   const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup)
{
    (void)sds; (void)sdd; (void)gdown; (void)gup;
    if (ns < 4) return -1;                                                               /* :88-91 */
    int smallest = (VLS(0, 1) < VLS(0, 2)) ? 1 : ns, largest = (VLS(0, 1) < VLS(0, 2)) ? ns : 1;
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            int f = AT(posn, i, t) - 1;                                                  /* :116 max(min(posn-1, ns-3), 1) */
            if (f > ns - 3) f = ns - 3;
            if (f < 1) f = 1;
            double x1 = VLS(i, f), x2 = VLS(i, f + 1), x3 = VLS(i, f + 2), x4 = VLS(i, f + 3);
            double z1 = SS(i, f), z2 = SS(i, f + 1), z3 = SS(i, f + 2), z4 = SS(i, f + 3);
            double cx1 = 1.0 / (x2 - x1), cx2 = 1.0 / (x3 - x1), cx3 = 1.0 / (x3 - x2);
            double cx4 = 1.0 / (x4 - x1), cx5 = 1.0 / (x4 - x2), cx6 = 1.0 / (x4 - x3);
            double a1 = z1;
            double a2 = cx1 * (z2 - z1);
            double a3 = cx2 * (cx3 * (z3 - z2) - cx1 * (z2 - z1));
            double a4 = cx4 * (cx5 * (cx6 * (z4 - z3) - cx3 * (z3 - z2)) - cx2 * (cx3 * (z3 - z2) - cx1 * (z2 - z1)));
            double xt = AT(vld, i, t);
            AT(sd, i, t) = (REAL)(a1 + (xt - x1) * (a2 + (xt - x2) * (a3 + a4 * (xt - x3))));       /* :144 */
            CLAMP_TAIL(i, t, 0)
        }
    return 0;
}

/* Interp1D_CubicWithDerivs_Body.inc:131-200: REAL(double) arithmetic, state and derivative */
int FN(orc_interp1d_cubicwithderivs)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                                     const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup)
{
    (void)gdown; (void)gup;
    if (ns < 2) return -1;                                                               /* :131-134 */
    int smallest = (VLS(0, 1) < VLS(0, 2)) ? 1 : ns, largest = (VLS(0, 1) < VLS(0, 2)) ? ns : 1;
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            int p = AT(posn, i, t);
            double lb = VLS(i, p), la = VLS(i, p + 1);
            double lc = 0.5 * (lb + la);
            double sb0 = SS(i, p), sa0 = SS(i, p + 1), sb1 = SDS(i, p), sa1 = SDS(i, p + 1);
            double dlin2 = la - lb;
            double o2 = 1 / dlin2;
            double dltar = (double)AT(vld, i, t) - lc;
            double dd = 0.125 * dlin2 * dlin2 - 0.5 * dltar * dltar;                      /* :173 */
            double sc0 = 0.5 * (sa0 + sb0);
            double sc1 = (sa0 - sb0) * o2;
            double sc2 = (sa1 - sb1) * o2;
            double sc3 = ((sa1 - sc1) - (sc1 - sb1)) * o2 * o2;
            double st2 = sc2 + 2. * dltar * sc3;
            double st1 = sc1 + dltar * st2;
            double st0 = sc0 + dltar * sc1;
            AT(sd, i, t) = (REAL)(st0 - dd * st2);                                       /* :187 */
            AT(sdd, i, t) = (REAL)(st1 - 2. * dd * sc3);                                 /* :188 */
            CLAMP_TAIL(i, t, 1)
        }
    return 0;
}

/* the level bookkeeping the three extrapolators share (Extrap1D_Fixed_Body.inc:62-75) */
#define EXTRAP_HEAD                                                                                 \
    int asc = VLS(0, 2) > VLS(0, 1);                                                               \
    int lowest = asc ? 1 : ns, highest = asc ? ns : 1, hiposn = asc ? ns - 1 : 1, loposn = asc ? 1 : ns - 1;

/* Extrap1D_Fixed_Body.inc:80-99: untouched everywhere else */
int FN(orc_extrap1d_fixed)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                           const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup)
{
    (void)ss; (void)sds; (void)sdd;
    EXTRAP_HEAD
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            if (xdown && AT(posn, i, t) == loposn) {
                if (AT(vld, i, t) < VLS(i, lowest)) AT(sd, i, t) = gdown;
            } else if (xup && AT(posn, i, t) == hiposn) {
                if (AT(vld, i, t) > VLS(i, highest)) AT(sd, i, t) = gup;
            }
        }
    return 0;
}

/* Extrap1D_LapseRate_Body.inc:92-116 */
int FN(orc_extrap1d_lapserate)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                               const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup)
{
    (void)sds; (void)sdd;
    EXTRAP_HEAD
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            if (xdown && AT(posn, i, t) == loposn) {
                if (AT(vld, i, t) < VLS(i, lowest)) {
                    REAL d = AT(vld, i, t) - VLS(i, lowest);
                    REAL pr = gdown * d;
                    AT(sd, i, t) = SS(i, lowest) + pr;                                   /* :98-101 */
                }
            } else if (xup && AT(posn, i, t) == hiposn) {
                if (AT(vld, i, t) > VLS(i, highest)) {
                    REAL d = AT(vld, i, t) - VLS(i, highest);
                    REAL pr = gup * d;
                    AT(sd, i, t) = SS(i, highest) + pr;                                  /* :107-110 */
                }
            }
        }
    return 0;
}

/* Extrap1D_Abort_Body.inc:70-92: the reference logs and calls exit(2) at the first offending point (loop order
 * vt outer, i inner); the oracle returns 2 and reports that point (1-based i, vt; 0 below / 1 above) */
int FN(orc_extrap1d_abort)(int n, int ns, int nd, int sij, int dij, const REAL *vls, const REAL *ss, const REAL *sds,
                           const int *posn, const REAL *vld, REAL *sd, REAL *sdd, int xdown, int xup, REAL gdown, REAL gup, int *where3)
{
    (void)ss; (void)sds; (void)sd; (void)sdd; (void)gdown; (void)gup;
    EXTRAP_HEAD
    for (int t = 0; t < nd; t++)
        for (int i = 0; i < n; i++) {
            if (xdown && AT(posn, i, t) == loposn) {
                if (AT(vld, i, t) < VLS(i, lowest)) { if (where3) { where3[0] = i + 1; where3[1] = t + 1; where3[2] = 0; } return 2; }
            } else if (xup && AT(posn, i, t) == hiposn) {
                if (AT(vld, i, t) > VLS(i, highest)) { if (where3) { where3[0] = i + 1; where3[1] = t + 1; where3[2] = 1; } return 2; }
            }
        }
    return 0;
}

#undef VLS
#undef SS
#undef SDS
#undef AT
#undef CLAMP_TAIL
#undef EXTRAP_HEAD
