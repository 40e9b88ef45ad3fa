/* emme_params.h -- plain-old-data parameter block shared by the C-ABI (emme_hip.h),
 * the host layer and the test oracle.
 *
 * Mirrors the data members of the reference's `struct Parameters` / `struct Stellarator`
 * (reference include/Parameters.h:14-42, 73-82) plus `iteration_precision`
 * (read in src/main.cpp:23).  Raw fields come from the input JSON; derived fields
 * (alpha, omega_s_i, ... curvature_aver) are filled by emme_params_derive()
 * following src/Parameters.cpp:36-66 and 211-223.
 */
#ifndef EMME_PARAMS_H
#define EMME_PARAMS_H

#ifdef __cplusplus
extern "C" {
#endif

/* `conf` values, reference src/Parameters.cpp:18-31 */
enum {
    EMME_CONF_TOKAMAK = 0,      /* "tokamak"            */
    EMME_CONF_STELLARATOR = 1,  /* "stellarator"        */
    EMME_CONF_CYLINDER = 2,     /* "cylinder"           */
    EMME_CONF_TAYLOR_MD = 3,    /* "taloyMagneticDrift" (sic) */
    EMME_CONF_CYLINDER_OLD = 4  /* "cylinder old"       */
};

/* `iteration_method`, reference src/main.cpp:41-49 */
enum { EMME_METHOD_TRACE_SECANT = 0, EMME_METHOD_QR_SECANT = 1 };

typedef struct emme_params {
    int conf;
    int iteration_method;
    /* raw JSON scalars (src/Parameters.cpp:37-66) */
    double q, shat, tau, epsilon_n, epsilon_r, eta_i, eta_e;
    double k_rho;
    double beta_e, R, vt, omega_d_coeff, length, theta;
    int npoints;
    int iteration_step_limit;
    double integration_precision; /* -> global_rel_tol  (functions.h:305 `tol`)  */
    double integration_accuracy;  /* -> precision_goal  (functions.h:305 `prec`) */
    int integration_iteration_limit;
    int integration_start_points; /* 15 or 31 */
    double arc_coeff;
    double water_bag_weight_vpara, water_bag_weight_vperp;
    int drift_center_transformation_switch;
    double iteration_precision; /* src/main.cpp:23 */
    double initial_guess[2];    /* src/main.cpp:205-206 */
    /* stellarator extras (src/Parameters.cpp:213-218) */
    double eta_k;
    int lh, mh;
    double epsilon_h_t, alpha_0, r_over_R;
    /* derived (filled by *_derive) */
    double b_theta; /* k_rho^2 */
    double alpha, omega_s_i, omega_s_e, omega_d_bar;
    double deltap, beta_e_p, rdeltapp, curvature_aver; /* stellarator */
    double shat_coeff;                                 /* cylinder    */
} emme_params_t;

#ifdef __cplusplus
}
#endif
#endif /* EMME_PARAMS_H */
