/* host_k0.h -- host-side model preparation (see host_k0.c). */
#ifndef HOST_K0_H
#define HOST_K0_H

#ifdef __cplusplus
extern "C" {
#endif

/* numbered as enum rate_mixture_mode of the reference (src/rate_mixture.h:11-17) */
enum {
    K0_MIX_NONE = 1,
    K0_MIX_CUSTOM = 2,       /* rates[] + prior[] as given */
    K0_MIX_UNIFORM = 3,      /* rates[] with prior 1/n */
    K0_MIX_GAMMA = 4,        /* mean rates of n equiprobable gamma classes (+I) */
    K0_MIX_GAMMA_MEDIAN = 5  /* normalised class medians (+I) */
};

typedef struct {
    int mode;
    int n;                   /* custom: length of rates/prior; gamma: gamma_categories */
    const double *rates;
    const double *prior;
    double gamma_shape;
    double invariable_prior;
} k0_mixture;

int arbplf_k0_category_count(const k0_mixture *mix);

/*
 * rate_matrix[k*k] raw (diagonal ignored).  Outputs: cat_rates[C], cat_prior[C],
 * pi_out[k] (zeros unless need_equilibrium), Qn_out[k*k] normalised with diagonal,
 * Qn_lo_out[k*k] (may be NULL) its low-order double-double words.
 * Returns C (>= 1) or -1 on allocation failure.
 */
int arbplf_k0_prepare(int k, const double *rate_matrix,
                      int use_equilibrium_divisor, double divisor_value, int need_equilibrium,
                      const k0_mixture *mix,
                      double *cat_rates, double *cat_prior, double *pi_out, double *Qn_out, double *Qn_lo_out);

#ifdef __cplusplus
}
#endif
#endif
