// Compile-time check (tests/test_abi.py): the hand-written RCCL declarations of metricsfm_amd/csrc/rccl_iface.h against the
// real /opt/rocm/include/rccl/rccl.h.  Nothing here runs; the translation unit only has to compile.
#include <rccl/rccl.h>
#include <type_traits>
#include "../include/msfm.h"
#include "../metricsfm_amd/csrc/rccl_iface.h"

static_assert(NCCL_UNIQUE_ID_BYTES == MSFM_RCCL_ID_BYTES && NCCL_UNIQUE_ID_BYTES == 128, "ncclUniqueId size");
static_assert(sizeof(ncclUniqueId) == sizeof(msfm_rccl::UniqueId) && alignof(ncclUniqueId) == alignof(msfm_rccl::UniqueId), "ncclUniqueId layout");
static_assert((int)ncclFloat64 == msfm_rccl::FLOAT64 && (int)ncclDouble == msfm_rccl::FLOAT64, "ncclFloat64");
static_assert((int)ncclSum == msfm_rccl::SUM, "ncclSum");
static_assert((int)ncclMax == msfm_rccl::MAX, "ncclMax");
static_assert((int)ncclSuccess == msfm_rccl::SUCCESS, "ncclSuccess");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int), "enum width");
static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclComm_t) == sizeof(void*), "ncclComm_t is a pointer");
static_assert(std::is_pointer<hipStream_t>::value && sizeof(hipStream_t) == sizeof(msfm_rccl::Stream), "hipStream_t is a pointer");

// the functions: same arity, and every parameter of the same size, alignment and kind (pointer / integer / aggregate) as ours
template <class A, class B>
constexpr bool same_abi() {
  return sizeof(A) == sizeof(B) && alignof(A) == alignof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value &&
         (std::is_integral<A>::value || std::is_enum<A>::value) == (std::is_integral<B>::value || std::is_enum<B>::value) &&
         std::is_class<A>::value == std::is_class<B>::value;
}
template <class F, class G> struct abi_match : std::false_type {};
template <class R1, class... A1, class R2, class... A2>
struct abi_match<R1 (*)(A1...), R2 (*)(A2...)> {
  static constexpr bool arity = sizeof...(A1) == sizeof...(A2);
  template <bool ok, class = void> struct each { static constexpr bool value = false; };
  template <class D> struct each<true, D> { static constexpr bool value = (same_abi<R1, R2>() && ... && same_abi<A1, A2>()); };
  static constexpr bool value = each<arity>::value;
};
static_assert(abi_match<decltype(&ncclGetUniqueId), msfm_rccl::get_unique_id_t>::value, "ncclGetUniqueId");
static_assert(abi_match<decltype(&ncclCommInitRank), msfm_rccl::comm_init_rank_t>::value, "ncclCommInitRank");
static_assert(abi_match<decltype(&ncclCommInitAll), msfm_rccl::comm_init_all_t>::value, "ncclCommInitAll");
static_assert(abi_match<decltype(&ncclAllReduce), msfm_rccl::all_reduce_t>::value, "ncclAllReduce");
static_assert(abi_match<decltype(&ncclCommDestroy), msfm_rccl::comm_destroy_t>::value, "ncclCommDestroy");
static_assert(abi_match<decltype(&ncclCommAbort), msfm_rccl::comm_abort_t>::value, "ncclCommAbort");
static_assert(abi_match<decltype(&ncclGroupStart), msfm_rccl::group_start_t>::value, "ncclGroupStart");
static_assert(abi_match<decltype(&ncclGroupEnd), msfm_rccl::group_end_t>::value, "ncclGroupEnd");
static_assert(abi_match<decltype(&ncclGetErrorString), msfm_rccl::error_string_t>::value, "ncclGetErrorString");
int main() { return 0; }
