def bytes_to_int(x: bytes) -> int:
    o = 0
    for b in x:
        o = (o << 8) + b
    return o
