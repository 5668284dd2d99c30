/* A C caller of librmn's EZ interpolation + packers as an application would write it: built against the headers of include/ and linked with -lrmn_ez_hip
 * (tests/callers/Makefile), no Python in the process.  The calls are the reference's own (ezscint.h:11-187, packers.h:4-15, armn_compress.h:17):
 *     c_ezqkdef x 2, c_ezsetopt, c_ezdefset, c_ezsint, c_ezuvint on host arrays; compact_float (16-bit slots, header style of c_fstecr) and
 *     armn_compress on the interpolated field -- the cfg5 chain of c_fstecr (fstd98.c:1170-1172).
 * usage: ez_c <in.bin> <out.bin>      in.bin = int32 ni nj no mo, float z[ni*nj] u[ni*nj] v[ni*nj]
 * out.bin = float zout[no*mo] uout[no*mo] vout[no*mo], int32 zlng, uint32 record[4 + no*mo/2 + 16] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ezscint_hip.h"
#include "packers_hip.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ez_c in.bin out.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    int32_t d[4];
    if (!f || fread(d, 4, 4, f) != 4) { fprintf(stderr, "ez_c: cannot read %s\n", argv[1]); return 2; }
    const int ni = d[0], nj = d[1], no = d[2], mo = d[3];
    const size_t nin = (size_t)ni * nj, nout = (size_t)no * mo;
    float *z = malloc(4 * nin), *u = malloc(4 * nin), *v = malloc(4 * nin);
    float *zo = malloc(4 * nout), *uo = malloc(4 * nout), *vo = malloc(4 * nout);
    if (fread(z, 4, nin, f) != nin || fread(u, 4, nin, f) != nin || fread(v, 4, nin, f) != nin) { fprintf(stderr, "ez_c: short input\n"); return 2; }
    fclose(f);
    char G[] = "G", L[] = "L", deg[] = "interp_degree", cubic[] = "cubic";
    const int32_t gdin = c_ezqkdef(ni, nj, G, 0, 0, 0, 0, 0);
    const int32_t gdout = c_ezqkdef(no, mo, L, 200, 200, 0, 0, 0);
    if (gdin < 0 || gdout < 0) { fprintf(stderr, "ez_c: c_ezqkdef failed\n"); return 1; }
    if (c_ezsetopt(deg, cubic) != 0) return 1;
    if (c_ezdefset(gdout, gdin) < 0) return 1;
    if (c_ezsint(zo, z) < 0) { fprintf(stderr, "ez_c: c_ezsint failed\n"); return 1; }
    if (c_ezuvint(uo, vo, u, v) < 0) { fprintf(stderr, "ez_c: c_ezuvint failed\n"); return 1; }
    /* c_fstecr's datyp 129 chain on the interpolated field: compact_float with 16-bit slots (nbits + 64 * 16), then armn_compress in place */
    const size_t rw = 4 + nout / 2 + 16;
    uint32_t *rec = calloc(rw, 4);
    double tempfloat = 99999.0;
    if (!compact_float(zo, rec, rec + 4, (int)nout, 16 + 64 * 16, 0, 1, 1, 0, &tempfloat)) { fprintf(stderr, "ez_c: compact_float failed\n"); return 1; }
    const int32_t zlng = armn_compress((unsigned char *)(rec + 4), no, mo, 1, 16, 1);
    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(zo, 4, nout, f); fwrite(uo, 4, nout, f); fwrite(vo, 4, nout, f);
    fwrite(&zlng, 4, 1, f); fwrite(rec, 4, rw, f);
    fclose(f);
    printf("ez_c: gdin %d gdout %d zlng %d z[0] %.6f z[last] %.6f\n", gdin, gdout, zlng, zo[0], zo[nout - 1]);
    return 0;
}
