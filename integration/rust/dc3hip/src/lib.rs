//! MI355X-native DC3 suffix sorting behind the same two functions every SACA crate of the
//! stringsearch workspace exposes (compare crates/cdivsufsort/src/lib.rs:1-30).
extern "C" {
    // include/dc3hip.h — same signature and return convention as divsufsort()
    fn dc3hip_sufsort_i32(T: *const u8, SA: *mut i32, n: i32) -> i32;
    fn dc3hip_sufsort_i64(T: *const u8, SA: *mut i64, n: i64) -> i32;
}

/// Sort suffixes of `text` and store their lexographic order in the given suffix array `sa`.
/// Will panic if `sa.len()` != `text.len()`
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

/// Sort suffixes
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
