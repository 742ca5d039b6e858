"""sacapart partition arithmetic (crates/sacapart/src/lib.rs:39-58) shared by the host mirror,
bench.py and the multi-process tests.  Pure host logic, no device code."""


def partition_size(text_len: int, num_partitions: int) -> int:
    """lib.rs:43: `text.len() / num_partitions + 1`."""
    if num_partitions < 1:
        raise ValueError("num_partitions must be >= 1")
    return text_len // num_partitions + 1


def chunk_bounds(text_len: int, num_partitions: int):
    """par_chunks(partition_size) of lib.rs:45-46: list of (offset, length); may hold fewer than
    num_partitions chunks (e.g. len 5, P 4 -> size 2 -> 3 chunks), never an empty chunk."""
    S = partition_size(text_len, num_partitions)
    return [(off, min(S, text_len - off)) for off in range(0, text_len, S)]


def rank_chunk(text_len: int, world_size: int, rank: int):
    """(offset, length) of the chunk rank `rank` owns when one process per GPU runs one chunk each;
    length 0 if par_chunks produced fewer chunks than ranks."""
    b = chunk_bounds(text_len, world_size)
    return b[rank] if rank < len(b) else (text_len, 0)
