/* Build-recipe header for oracle/_ref (TEST INFRASTRUCTURE, not product code).
 *
 * The reference generates nlopt_config.h with cmake (CMakeLists.txt:135 from
 * nlopt_config.h.in).  We do not run the reference's build system; this is the
 * hand-written equivalent for the one platform we build on (x86-64 Linux,
 * gcc 11, glibc 2.35).  Every value below is what cmake's probes produce on
 * this image (checked against SURVEY.md §8c's cmake build: 77/77 ctest pass).
 */
#ifndef ORACLE_REF_NLOPT_CONFIG_H
#define ORACLE_REF_NLOPT_CONFIG_H
#define MAJOR_VERSION 2
#define MINOR_VERSION 11
#define BUGFIX_VERSION 0
#define HAVE_COPYSIGN
#define HAVE_FPCLASSIFY
#define HAVE_GETOPT_H
#define HAVE_GETOPT
#define HAVE_GETPID
#define HAVE_GETTIMEOFDAY
#define HAVE_ISINF
#define HAVE_ISNAN
#define HAVE_QSORT_R
#define HAVE_STDINT_H
#define HAVE_SYS_TIME_H
#define HAVE_TIME
#define HAVE_UINT32_T
#define HAVE_UNISTD_H
#define SIZEOF_UNSIGNED_INT 4
#define SIZEOF_UNSIGNED_LONG 8
#define THREADLOCAL __thread
#define TIME_WITH_SYS_TIME
/* NLOPT_CXX intentionally undefined: stogo/ags (C++) are off the hot path. */
#endif
