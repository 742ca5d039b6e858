//! MI355X-native DC3 suffix sorting behind the same two functions every SACA crate of the
//! stringsearch workspace exposes (compare crates/cdivsufsort/src/lib.rs:1-30).
extern "C" {
    // include/dc3hip.h — same signature and return convention as divsufsort()
    fn dc3hip_sufsort_i32(T: *const u8, SA: *mut i32, n: i32) -> i32;
    fn dc3hip_sufsort_i64(T: *const u8, SA: *mut i64, n: i64) -> i32;
    fn dc3hip_sufsort_ex(T: *const u8, SA: *mut core::ffi::c_void, n: i64, opts: *const Dc3hipOpts) -> i32;
}

/// `dc3hip_opts` of include/dc3hip.h
#[repr(C)]
struct Dc3hipOpts {
    struct_size: i32,
    index_bits: i32,
    device: i32,
    num_partitions: i32,
    flags: i32,
}
const DC3HIP_F_ALL_DEVICES: i32 = 2;

/// Builds the suffix array of `text` on the GPU into the caller's `sa` (one i32 per byte of `text`).
/// Same contract as the workspace's other SACA crates: the two slices must have equal lengths and the
/// text must be addressable with i32 indices; a library error (no device, out of HBM) panics, as the
/// reference's `assert_eq!(0, ret)` does.
pub fn sort_in_place(text: &[u8], sa: &mut [i32]) {
    assert_eq!(text.len(), sa.len(), "text and suffix array should have same len");
    assert!(
        text.len() < i32::max_value() as usize,
        "text too large, should not exceed {} bytes",
        i32::max_value() - 1
    );
    let ret = unsafe { dc3hip_sufsort_i32(text.as_ptr(), sa.as_mut_ptr(), text.len() as i32) };
    assert_eq!(0, ret);
}

/// Allocating variant: returns a `sacabase::SuffixArray` that borrows `text` and owns the index array.
pub fn sort<'a>(text: &'a [u8]) -> sacabase::SuffixArray<'a, i32> {
    let mut sa = vec![0; text.len()];
    sort_in_place(text, &mut sa);
    sacabase::SuffixArray::new(text, sa)
}

/// 64-bit indices (sacabase only needs `Index: ToPrimitive`)
pub fn sort_i64<'a>(text: &'a [u8]) -> sacabase::SuffixArray<'a, i64> {
    let mut sa = vec![0i64; text.len()];
    let ret = unsafe { dc3hip_sufsort_i64(text.as_ptr(), sa.as_mut_ptr(), text.len() as i64) };
    assert_eq!(0, ret);
    sacabase::SuffixArray::new(text, sa)
}

/// The local suffix arrays of sacapart's chunks (`len / num_partitions + 1` bytes each,
/// crates/sacapart/src/lib.rs:43-46) from one library call, back to back; the node's GPUs share the
/// chunks (one host worker per GPU), which is what `par_chunks` does with CPU cores (lib.rs:45-49).
pub fn sort_partitions(text: &[u8], num_partitions: u32) -> Vec<i32> {
    let mut sa = vec![0i32; text.len()];
    if text.is_empty() {
        return sa;
    }
    let opts = Dc3hipOpts {
        struct_size: core::mem::size_of::<Dc3hipOpts>() as i32,
        index_bits: 32,
        device: -1,
        num_partitions: num_partitions as i32,
        flags: DC3HIP_F_ALL_DEVICES,
    };
    let ret = unsafe { dc3hip_sufsort_ex(text.as_ptr(), sa.as_mut_ptr() as *mut _, text.len() as i64, &opts) };
    assert_eq!(0, ret);
    sa
}

#[cfg(test)]
mod tests {
    // same corpus test as crates/divsufsort/src/lib.rs:83-91
    #[test]
    fn shruggy() {
        let input = "¯\\_(ツ)_/¯".as_bytes();
        let sa = super::sort(input);
        sa.verify().unwrap();
    }
}
