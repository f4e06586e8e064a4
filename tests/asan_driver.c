/* Host-side AddressSanitizer driver (tests/test_asan_cpu.py builds libgsmcal.so with -fsanitize=address -fno-gpu-sanitize
 * and runs this): the entry points that need no GPU, and the failure paths a GPU-less machine takes. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "gsmcal.h"
int main(void) {
    gsmcal_params p; gsmcal_params_default(&p);
    double in[2] = {34.78, -1.08}, out = 0;
    int rc = gsmcal_total_ppm_calculation(in, 2, &out);
    double inf2[2] = {INFINITY, INFINITY};
    int rc2 = gsmcal_total_ppm_calculation(inf2, 2, &out);
    gsmcal_ctx* c = NULL;
    int rc3 = gsmcal_ctx_create(0, &c);      /* no GPU here: must fail cleanly */
    unsigned char id[GSMCAL_COMM_ID_BYTES];
    int rc4 = gsmcal_comm_get_unique_id(id);
    printf("th %.1f rc %d rc2 %d ctx %d (%p) id %d version %s\n", p.coarse_th_db, rc, rc2, rc3, (void*)c, rc4, gsmcal_version());
    return (rc == 0 && rc2 == GSMCAL_S_ALL_INF && rc3 < 0 && c == NULL) ? 0 : 1;
}
