// The handful of libza_amd.so internals the scaffolding library (libza_amd_synth.so: synthetic workloads, self-test hooks)
// needs.  libza_amd.so exports the C functions of include/za_amd.h and ONE data symbol, this table -- no C++ symbol
// leaves it (csrc/libza_amd.map) -- so a Rust / cgo / C++ host linking the product cannot collide with, or come to depend
// on, its internals.  Not part of the boundary: the layout may change with any build; both libraries are built together.
#pragma once
#include "common.h"
#include "formats.h"

struct zg_internal_api {
    int (*fail_msg)(zg::Ctx* ctx, int code, const char* msg);
    int (*on_exception)(zg::Ctx* ctx);
    int (*convert_mont_dev)(zg::Ctx* ctx, void* d_elems, size_t n, int which, int dir, hipStream_t st);
    int (*host_pack_for_test)(const uint8_t* witness32, size_t nv, int threads, uint8_t* classes, uint32_t* other_index,
                              uint8_t* others32, size_t* n_others);
    int (*host_pack_split_for_test)(const uint8_t* witness32, size_t nv, int threads, int cap_permille, int ready_every,
                                    uint8_t* raw_out, uint8_t* classes, uint32_t* other_index, uint8_t* others32, size_t* n_others,
                                    size_t* first);
    int (*params_h_eval_slice)(zg::Ctx* ctx, const zg_params* p, const zg::G1Affine* d_h_full, uint64_t n_h,
                               const zg::G1Affine* d_h_eval_full);
    int (*parse_bellman_params)(zg::Ctx* ctx, const uint8_t* buf, size_t len, bool checked, zg::ParsedParams* out, size_t* consumed);
    void (*write_bellman_params)(const zg::ParsedParams& p, std::vector<uint8_t>* out);
    bool (*pairing_product_is_one)(const zg::G2Affine* qs, const zg::G1Affine* ps, int n, bool slow, uint8_t* gt_out);
};
extern "C" const zg_internal_api zg_internal_v1;
