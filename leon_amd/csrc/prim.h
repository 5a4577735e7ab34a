// prim.h -- the device-wide primitives the path uses (exclusive sum, radix sort, reduce, flagged select): rocPRIM called
// directly.  Same argument order at the call sites as before: (temporary storage, its size, ..., stream); a null
// temporary-storage pointer only returns the size.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>

#include <iterator>

namespace prim {

template <typename In, typename Out>
inline hipError_t ExclusiveSum(void* tmp, size_t& bytes, In in, Out out, size_t n, hipStream_t s) {
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::exclusive_scan(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}
template <typename In, typename Out>
inline hipError_t Sum(void* tmp, size_t& bytes, In in, Out out, size_t n, hipStream_t s) {
    using T = typename std::iterator_traits<Out>::value_type;
    return rocprim::reduce(tmp, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}
template <typename Key, typename Val>
inline hipError_t SortPairs(void* tmp, size_t& bytes, const Key* kin, Key* kout, const Val* vin, Val* vout, size_t n, unsigned begin_bit,
                            unsigned end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}
template <typename Key>
inline hipError_t SortKeys(void* tmp, size_t& bytes, const Key* kin, Key* kout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t s) {
    return rocprim::radix_sort_keys(tmp, bytes, kin, kout, n, begin_bit, end_bit, s);
}
template <typename In, typename Flag, typename Out, typename Count>
inline hipError_t Flagged(void* tmp, size_t& bytes, In in, Flag flags, Out out, Count n_selected, size_t n, hipStream_t s) {
    return rocprim::select(tmp, bytes, in, flags, out, n_selected, n, s);
}

}  // namespace prim
