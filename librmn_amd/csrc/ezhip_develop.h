/* Development switches.  The shipped library (make) is built WITHOUT -DEZHIP_DEVELOP: every knock-out that changes what a kernel
 * computes or stores (EZHIP_DEBUG, EZHIP_ENC_DEBUG, EZHIP_CFG5_ONLY_FUSED, EZHIP_SEPX_PAD ...) is then compiled out -- the kernels
 * see the constant 0 and the host never reads those environment variables.  `make develop` builds devlibs/librmn_ez_hip_dev.so with
 * them in, for the measurement scripts under tools/ (EZHIP_LIBRARY=devlibs/librmn_ez_hip_dev.so).  ezhip_develop_build() tells which one
 * a process loaded; bench.py records it in its JSON line.
 *
 * Environment variables that only choose between routes producing the SAME results (EZHIP_NO_SEPX, EZHIP_FORCE_PTS, ... -- the tests
 * use them to force each kernel family through the same parity checks) stay in the shipped build; bench.py lists every EZHIP_* variable
 * set in its environment (config.env_overrides). */
#ifndef EZHIP_DEVELOP_H
#define EZHIP_DEVELOP_H
#ifdef EZHIP_DEVELOP
#define EZH_DEVENV(name) getenv(name)
#define EZH_DBG(x) (x)
#define EZH_DEVINT(name) (getenv(name) ? atoi(getenv(name)) : 0)
#define EZH_DEVELOP_BUILD 1
#else
#define EZH_DEVENV(name) ((const char *)0)
#define EZH_DBG(x) 0
#define EZH_DEVINT(name) 0
#define EZH_DEVELOP_BUILD 0
#endif
#endif
