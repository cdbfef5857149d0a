// tools/probes/plan_check.c -- the phase plan of seeds.hpp (host build: csdr_amd_debug_phase_chain) against the loop of libcsdr_gpl.c:48-51, bit for bit, on
// <rates> x <chunks> values (default 2000 x 24000 = 48 M; rates random in +-0.5 plus the binade edges).  Build: gcc -O1 -ffp-contract=off -o /tmp/plan_check
// tools/probes/plan_check.c -Lcsdr_amd -lcsdr_amd -Wl,-rpath,$PWD/csdr_amd -Wl,-rpath,/opt/rocm/lib
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
void csdr_amd_debug_phase_chain(float rate, float ph0, int n, float *out);
static float loopw(float x) { const float pi = (float)3.14159265358979323846; const float c = 2 * pi; while (x > pi) x = x - c; while (x < -pi) x = x + c; return x; }
int main(int argc, char **argv)
{
    int nr = argc > 1 ? atoi(argv[1]) : 2000, n = argc > 2 ? atoi(argv[2]) : 24000;
    float *out = malloc(sizeof(float) * n);
    uint64_t seed = 12345; long bad = 0, total = 0;
    for (int i = 0; i < nr; i++) {
        seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
        double u = (double)(seed >> 11) / 9007199254740992.0;
        float rate = (float)(u - 0.5);
        if (i % 50 == 0) { const double e[] = {16, 32, 64, 256, 512, 1024, 2048}; rate = (float)(e[(i / 50) % 7] / (2 * 3.14159265358979323846 * 1024) * ((i / 350) % 2 ? -1 : 1)) * (1.0f + ((i / 700) % 3 - 1) * 1e-6f); }
        float ph = (i & 1) ? (float)((u * 7.0 - (int)(u * 7.0)) * 6.28 - 3.14) : 0.f;
        csdr_amd_debug_phase_chain(rate, ph, n, out);
        const float pi = (float)3.14159265358979323846;
        volatile float inc = (rate * 2) * pi; volatile float step = inc * (float)1024;
        for (int k = 0; k < n; k++) { volatile float x = ph + step; ph = loopw(x); uint32_t a, b; memcpy(&a, &ph, 4); memcpy(&b, &out[k], 4); if (a != b) { if (bad < 5) printf("rate %g k %d: %a vs %a\n", rate, k, ph, out[k]); bad++; } total++; }
    }
    printf("%ld values, %ld differ\n", total, bad);
    return bad != 0;
}
