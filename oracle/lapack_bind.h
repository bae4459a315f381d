// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into or called from the product path.
//
// Run-time binding to a host LAPACK (LP64, Fortran ABI).  The oracle restates the reference's CPU
// algorithm as the same sequence of BLAS/LAPACK calls the reference makes through BLAS++/LAPACK++
// (RandLAPACK/rl_blaspp.hh:3, RandLAPACK/rl_lapackpp.hh:5); those two libraries are thin C++ wrappers
// over exactly these Fortran symbols, and neither they nor RandBLAS nor Random123 are present in this
// image, so the reference itself is unbuildable here.
// The library is located at run time (scipy's bundled OpenBLAS, symbols prefixed `scipy_`, or any
// system liblapack/libopenblas with plain names) so the same liboracle.so works on the GPU box.
#pragma once
#include <cstdint>
#include <cstddef>

namespace orc {

typedef int lint;  // LP64 Fortran INTEGER

struct Lapack {
    void* handle = nullptr;
    const char* prefix = "";
    // BLAS
    void (*dgemm)(const char*, const char*, const lint*, const lint*, const lint*, const double*, const double*,
                  const lint*, const double*, const lint*, const double*, double*, const lint*, size_t, size_t);
    void (*dsyrk)(const char*, const char*, const lint*, const lint*, const double*, const double*, const lint*,
                  const double*, double*, const lint*, size_t, size_t);
    void (*dtrsm)(const char*, const char*, const char*, const char*, const lint*, const lint*, const double*,
                  const double*, const lint*, double*, const lint*, size_t, size_t, size_t, size_t);
    void (*dtrmm)(const char*, const char*, const char*, const char*, const lint*, const lint*, const double*,
                  const double*, const lint*, double*, const lint*, size_t, size_t, size_t, size_t);
    void (*dger)(const lint*, const lint*, const double*, const double*, const lint*, const double*, const lint*,
                 double*, const lint*);
    void (*dscal)(const lint*, const double*, double*, const lint*);
    // LAPACK
    void (*dpotrf)(const char*, const lint*, double*, const lint*, lint*, size_t);
    void (*dgesdd)(const char*, const lint*, const lint*, double*, const lint*, double*, double*, const lint*,
                   double*, const lint*, double*, const lint*, lint*, lint*, size_t);
    void (*dgeqrf)(const lint*, const lint*, double*, const lint*, double*, double*, const lint*, lint*);
    void (*dorgqr)(const lint*, const lint*, const lint*, double*, const lint*, const double*, double*,
                   const lint*, lint*);
    void (*dormqr)(const char*, const char*, const lint*, const lint*, const lint*, const double*, const lint*,
                   const double*, double*, const lint*, double*, const lint*, lint*, size_t, size_t);
    void (*dgeqp3)(const lint*, const lint*, double*, const lint*, lint*, double*, double*, const lint*, lint*);
    void (*dgetrf)(const lint*, const lint*, double*, const lint*, lint*, lint*);
    void (*dlaswp)(const lint*, double*, const lint*, const lint*, const lint*, const lint*, const lint*);
    double (*dlange)(const char*, const lint*, const lint*, const double*, const lint*, double*, size_t);
    void (*dlacpy)(const char*, const lint*, const lint*, const double*, const lint*, double*, const lint*, size_t);
    void (*dlaset)(const char*, const lint*, const lint*, const double*, const double*, double*, const lint*,
                   size_t);
    void (*dlapmt)(const lint*, const lint*, const lint*, double*, const lint*, lint*);
    void (*dorhr_col)(const lint*, const lint*, const lint*, double*, const lint*, double*, const lint*, double*,
                      lint*);
    void (*dgeqrt)(const lint*, const lint*, const lint*, double*, const lint*, double*, const lint*, double*,
                   lint*);
    void (*dgemqrt)(const char*, const char*, const lint*, const lint*, const lint*, const lint*, const double*,
                    const lint*, const double*, const lint*, double*, const lint*, double*, lint*, size_t, size_t);
    void (*dlarfg)(const lint*, double*, double*, const lint*, double*);
    void (*dlarf)(const char*, const lint*, const lint*, const double*, const lint*, const double*, double*, const lint*,
                  double*, size_t);
    void (*dlarfb)(const char*, const char*, const char*, const char*, const lint*, const lint*, const lint*, const double*,
                   const lint*, const double*, const lint*, double*, const lint*, double*, const lint*, size_t, size_t, size_t,
                   size_t);
    void (*dlarft)(const char*, const char*, const lint*, const lint*, const double*, const lint*, const double*, double*,
                   const lint*, size_t, size_t);
    double (*dnrm2)(const lint*, const double*, const lint*);
    lint (*idamax)(const lint*, const double*, const lint*);
    void (*dswap)(const lint*, double*, const lint*, double*, const lint*);
    // single precision twins of the routines the BQRRP restatement calls (the fp32 leg of the oracle: BASELINE configs[3] is fp32)
    void (*sgemm)(const char*, const char*, const lint*, const lint*, const lint*, const float*, const float*, const lint*, const float*, const lint*,
                  const float*, float*, const lint*, size_t, size_t);
    void (*ssyrk)(const char*, const char*, const lint*, const lint*, const float*, const float*, const lint*, const float*, float*, const lint*, size_t,
                  size_t);
    void (*strsm)(const char*, const char*, const char*, const char*, const lint*, const lint*, const float*, const float*, const lint*, float*,
                  const lint*, size_t, size_t, size_t, size_t);
    void (*strmm)(const char*, const char*, const char*, const char*, const lint*, const lint*, const float*, const float*, const lint*, float*,
                  const lint*, size_t, size_t, size_t, size_t);
    void (*spotrf)(const char*, const lint*, float*, const lint*, lint*, size_t);
    void (*sgeqrf)(const lint*, const lint*, float*, const lint*, float*, float*, const lint*, lint*);
    void (*sormqr)(const char*, const char*, const lint*, const lint*, const lint*, const float*, const lint*, const float*, float*, const lint*, float*,
                   const lint*, lint*, size_t, size_t);
    void (*sgeqp3)(const lint*, const lint*, float*, const lint*, lint*, float*, float*, const lint*, lint*);
    void (*sgetrf)(const lint*, const lint*, float*, const lint*, lint*, lint*);
    void (*slacpy)(const char*, const lint*, const lint*, const float*, const lint*, float*, const lint*, size_t);
    void (*slaset)(const char*, const lint*, const lint*, const float*, const float*, float*, const lint*, size_t);
    void (*sorhr_col)(const lint*, const lint*, const lint*, float*, const lint*, float*, const lint*, float*, lint*);
    void (*sgeqrt)(const lint*, const lint*, const lint*, float*, const lint*, float*, const lint*, float*, lint*);
    void (*sgemqrt)(const char*, const char*, const lint*, const lint*, const lint*, const lint*, const float*, const lint*, const float*, const lint*,
                    float*, const lint*, float*, lint*, size_t, size_t);
    void (*sorgqr)(const lint*, const lint*, const lint*, float*, const lint*, const float*, float*, const lint*, lint*);
    void (*set_threads)(int);
    int (*get_threads)(void);
};

// returns nullptr on success, else a static error string
const char* lapack_open(const char* path);
Lapack& lapack();

}  // namespace orc
