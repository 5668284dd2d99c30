/* orc_interpv.h -- TEST INFRASTRUCTURE: the oracle's 1-D vertical interpolation (see orc_interpv.c). */
#ifndef ORC_INTERPV_H
#define ORC_INTERPV_H
#ifdef __cplusplus
extern "C" {
#endif
#define ORC_IV_DECL(R, S)                                                                                                    \
    void orc_interp1d_findpos##S(int n, int ns, int nd, int sij, int dij, const R *vls, int *posn, const R *vld);            \
    int orc_interp1d_nearestneighbour##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R); \
    int orc_interp1d_linear##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R);           \
    int orc_interp1d_cubiclagrange##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R);    \
    int orc_interp1d_cubicwithderivs##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R);  \
    int orc_extrap1d_fixed##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R);            \
    int orc_extrap1d_lapserate##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R);        \
    int orc_extrap1d_abort##S(int, int, int, int, int, const R *, const R *, const R *, const int *, const R *, R *, R *, int, int, R, R, int *where3);
ORC_IV_DECL(float, )
ORC_IV_DECL(double, 8)
#ifdef __cplusplus
}
#endif
#endif
