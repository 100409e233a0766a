// The few RCCL types and values libmsfm uses, declared by hand: rccl.h is not included by the library on purpose (libmsfm
// must build and load where RCCL is absent; librccl is opened with dlopen).  tests/rccl_iface_check.cpp includes this header
// TOGETHER with /opt/rocm/include/rccl/rccl.h and static_asserts that every declaration here agrees with the real one
// (tests/test_abi.py::test_rccl_interface_matches_rccl_h compiles it).
#pragma once
#include <cstddef>

#ifndef MSFM_RCCL_ID_BYTES
#define MSFM_RCCL_ID_BYTES 128
#endif

namespace msfm_rccl {
struct UniqueId { char internal[MSFM_RCCL_ID_BYTES]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Stream;                                      // hipStream_t
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*comm_init_all_t)(void** comms, int ndev, const int* devlist);
typedef int (*all_reduce_t)(const void* send, void* recv, size_t count, int datatype, int op, void* comm, Stream stream);
typedef int (*comm_destroy_t)(void* comm);
typedef int (*comm_abort_t)(void* comm);
typedef int (*group_start_t)();
typedef int (*group_end_t)();
typedef const char* (*error_string_t)(int);
enum { SUM = 0, MAX = 2, FLOAT64 = 8, SUCCESS = 0 };       // ncclRedOp_t / ncclDataType_t / ncclResult_t values of rccl.h
}  // namespace msfm_rccl
