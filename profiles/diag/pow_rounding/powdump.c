#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static double my_pow(double a, double b) { double r = pow(a, b); fprintf(stderr, "%a %a %a\n", a, b, r); return r; }
#define pow my_pow
#include "../../../oracle/gfir_interp.c"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    uint8_t *d = malloc(n); fread(d, 1, n, f); fclose(f);
    gfi_item *item = gfi_load(d, n);
    double state[8]; double *cols[8]; double out, *outs[1] = {&out};
    for (int i = 0; i < 8; i++) { state[i] = strtod(argv[2 + i], 0); cols[i] = &state[i]; }
    gfi_run_f64(item, cols, outs, 0, 1);
    for (int i = 0; i < 8; i++) printf("%a ", state[i]);
    printf("\n");
    return 0;
}
