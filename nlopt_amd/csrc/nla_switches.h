/* nla_switches.h — development switches.  The NLA_* environment variables (A/B switches of the benches, debug dumps, the stand-in
 * collective library of the multi-process CPU tests) exist only in builds with -DNLA_DEBUG_SWITCHES: the emulated-device test
 * library (oracle/Makefile) and the instrumented variants built by tools/ (NLOPT_AMD_VARIANT).  The shipped libnlopt_amd.so reads
 * no environment variable: behaviour is set through the API (nlopt_set_param "amd_*", include/nlopt_amd.h) only. */
#ifndef NLA_SWITCHES_H
#define NLA_SWITCHES_H
#include <stdlib.h>
#ifdef NLA_DEBUG_SWITCHES
#define NLA_DBG_ENV(name) getenv(name)
#else
#define NLA_DBG_ENV(name) ((const char *) 0)
#endif
/* integer value of a development switch, `dflt` where it is not set (always, in the shipped library) */
static inline int nla_dbg_int(const char *name, int dflt) { const char *e = NLA_DBG_ENV(name); (void) name; return e ? atoi(e) : dflt; }
#endif
