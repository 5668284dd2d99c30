/* oracle/orc_compact_tmpl.h -- TEST INFRASTRUCTURE (CPU oracle).  Body of compact_float / compact_double, included twice by
 * orc_pack.c with ORC_FT = float / double, the way src/packers/compact.c:27-37 includes compact.tmplc (:37-431) twice. */
void *ORC_COMPACT_NAME(void *unpacked, void *packedHeader, void *packed, int elementCount,
                        int packedTokenBitSize, int offset, int stride, int opCode, int hasMissing, const void *missingTag)
{
    ORC_FT *a = (ORC_FT *)unpacked;
    uint32_t *hdr = (uint32_t *)packedHeader, *out = (uint32_t *)packed;
    float missingValueTag = *(const ORC_FT *)missingTag;          /* a float in both instantiations (compact.tmplc:106) */
    if (packedTokenBitSize == 0) return NULL;
    if (packedTokenBitSize == 1 && hasMissing) return NULL;
    int bs = packedTokenBitSize, eff;
    if (bs > 64) { eff = bs >> 6; bs &= 0x3F; } else eff = bs;                     /* :121-129 */
    if (opCode == 1) {
        uint32_t n = (uint32_t)elementCount;
        uint32_t missingToken = (bs != 32) ? ~(0xFFFFFFFFu << bs) : ~0u;
        int style = ((&hdr[3] == out && offset == 24) || (&hdr[0] == out && offset == 120)) ? 1 : 2;   /* :159-168 */
        if (style == 2 && n > 268435455u) return NULL;
        uint32_t countLower20 = (n << 12) >> 12, countUpper8 = (n << 4) >> 24;
        double maxF, minF;
        if (!hasMissing) {                                                           /* :173-186 */
            maxF = minF = a[0];
            for (size_t i = (size_t)stride; i < (size_t)n * stride; i += stride) {
                if (a[i] < minF) minF = a[i]; else if (a[i] > maxF) maxF = a[i];
            }
        } else {                                                                     /* :187-204 */
            size_t i = 0;
            while (a[i] == missingValueTag) i += stride;
            maxF = minF = a[i];
            for (i = (size_t)stride; i < (size_t)n * stride; i += stride) {
                if (a[i] == missingValueTag) continue;
                if (a[i] < minF) minF = a[i]; else if (a[i] > maxF) maxF = a[i];
            }
        }
        if (maxF > 1.0e+38 || minF < -1.0e+38) { fprintf(stderr, "orc_compact: number too large\n"); exit(33); }
        dbits range, minT;
        range.d = (maxF - minF) * 2;
        minT.d = minF;
        range.u &= 0xFFF0000000000000ull;                                            /* mantissa := 0 (:212-214) */
        uint32_t tempInt = (range.d == 0) ? 0 : (uint32_t)(int64_t)((maxF - minF) * ldexp(1.0, bs) / range.d);
        if (tempInt == missingToken && hasMissing) range.u += 0x0010000000000000ull;  /* expo++ (:223-225) */
        int rexpo = (int)((range.u >> 52) & 0x7FF);
        int tempExpo = (range.d == 0) ? 0 : (rexpo - 1023);
        uint32_t scaledExpOfMinFloat = (uint32_t)((int)((minT.u >> 52) & 0x7FF) - 1023 + 1024 - 48);
        uint32_t scaledExpOfRange = (uint32_t)(tempExpo - bs);
        uint32_t signOfMinFloat = (minF < 0) ? 1 : 0;
        if (minF == 0.0) scaledExpOfMinFloat &= 0x00000111;                          /* sic, :240-242 */
        uint32_t headerType = (style == 1) ? (hasMissing == 1 ? 0x7ef : 0x7ff) : (hasMissing == 1 ? 0xfef : 0xfff);
        hdr[0] = headerType << 20 | countLower20;
        hdr[1] = ((scaledExpOfRange + 4096) << 16) | ((scaledExpOfMinFloat << 4) | signOfMinFloat);
        if (minF == 0.0) hdr[2] = 0;
        else {
            uint32_t m1 = (uint32_t)((minT.u >> 32) & 0xFFFFF), m2 = (uint32_t)((minT.u >> 29) & 0x7);
            hdr[2] = 0x80000000u | (m1 << 11) | (m2 << 8);
        }
        hdr[3] = (uint32_t)bs << 8 | countUpper8;
        double mulFactor = ldexp(1.0, bs) / ldexp(1.0, tempExpo);                    /* f_pow(2, tempExpo) */
        bitw w;
        bw_init(&w, out, offset);
        if (w.space == 32 && bs == 32) {                                             /* direct copy :302-312 */
            uint32_t *p = w.ptr;
            for (size_t i = 0; i < (size_t)n * stride; i += stride)
                *p++ = (hasMissing == 1 && a[i] == missingValueTag) ? missingToken : (uint32_t)((a[i] - minF) * mulFactor);
            return out;
        }
        for (size_t i = 0; i < (size_t)n * stride; i += stride) {
            uint32_t t = (hasMissing == 1 && a[i] == missingValueTag) ? missingToken
                                                                      : (uint32_t)(int64_t)(((double)a[i] - minF) * mulFactor);
            bw_put(&w, t, eff);
        }
        bw_flush(&w);
        return out;
    }
    if (opCode == 2) {                                                               /* FLOAT_UNPACK :336-425 */
        uint32_t marker = hdr[0] >> 20, counter = hdr[0] & 0xFFFFF;
        uint32_t rangeExpo = hdr[1] >> 16, minExpo = (hdr[1] >> 4) & 0xFFF, minSign = hdr[1] & 0xF;
        uint32_t minMantisa32 = hdr[2], bitSize = (hdr[3] >> 8) & 0xFF, emptySpace = hdr[3] & 0xFF;
        uint32_t intCount = (marker == 0x7ff || marker == 0x7ef) ? (uint32_t)elementCount : (emptySpace << 20 | counter);
        int tokenSize = (int)bitSize;
        uint32_t missingToken = (tokenSize != 32) ? ~(0xFFFFFFFFu << tokenSize) : ~0u;
        uint32_t rangeExponent = rangeExpo - 4096 + 127 + tokenSize;
        double mulFactor = ldexp(1.0, (int)(rangeExponent - 127 - tokenSize));
        double minF;
        if (minMantisa32 == 0 || minExpo < 849) minF = 0;
        else {
            union { float f; uint32_t u; } m;
            m.u = (minSign & 1) << 31 | ((minExpo + 127 - 1024 + 48) & 0xFF) << 23 | ((minMantisa32 >> 8) & 0x7fffff);
            minF = m.f;
        }
        bitr r;
        br_init(&r, out, offset);
        int sig = tokenSize > 32 ? 32 : tokenSize, insig = tokenSize > 32 ? tokenSize - 32 : 0;
        for (size_t i = 0; i < (size_t)intCount * stride; i += stride) {
            uint32_t t = br_get(&r, sig);
            if (insig > 0) (void)br_get(&r, insig);
            if (hasMissing == 1 && t == missingToken) a[i] = missingValueTag;
            else if (t == 0) a[i] = (ORC_FT)minF;
            else a[i] = (ORC_FT)((t * mulFactor) * 1.0000000000001 + minF);
        }
        return a;
    }
    return NULL;
}

