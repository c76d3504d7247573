"""Wire-format constants of Nimble's cascading stream encodings.

Enum values follow dwio/nimble/common/Types.h:69-197 of the reference; the
physical type of every numeric data type is the same-width unsigned integer
(dwio/nimble/common/Types.h:253-321), floats travel as bit patterns.
"""
from __future__ import annotations

import numpy as np

# EncodingType (dwio/nimble/common/Types.h:69-163)
TRIVIAL = 0
RLE = 1
DICTIONARY = 2
FIXED_BIT_WIDTH = 3
SENTINEL = 4
NULLABLE = 5
SPARSE_BOOL = 6
VARINT = 7
DELTA = 8
CONSTANT = 9
MAINLY_CONSTANT = 10
BLOCK_BIT_PACKING = 15  # experimental in the reference
FOR = 18  # experimental in the reference

ENCODING_NAMES = {
    TRIVIAL: "Trivial",
    RLE: "RLE",
    DICTIONARY: "Dictionary",
    FIXED_BIT_WIDTH: "FixedBitWidth",
    SENTINEL: "Sentinel",
    NULLABLE: "Nullable",
    SPARSE_BOOL: "SparseBool",
    VARINT: "Varint",
    DELTA: "Delta",
    CONSTANT: "Constant",
    MAINLY_CONSTANT: "MainlyConstant",
    BLOCK_BIT_PACKING: "BlockBitPacking",
    FOR: "FOR",
}

# DataType (dwio/nimble/common/Types.h:169-183)
UNDEFINED = 0
INT8 = 1
UINT8 = 2
INT16 = 3
UINT16 = 4
INT32 = 5
UINT32 = 6
INT64 = 7
UINT64 = 8
FLOAT = 9
DOUBLE = 10
BOOL = 11
STRING = 12

DATA_TYPE_NAMES = {
    INT8: "int8", UINT8: "uint8", INT16: "int16", UINT16: "uint16",
    INT32: "int32", UINT32: "uint32", INT64: "int64", UINT64: "uint64",
    FLOAT: "float", DOUBLE: "double", BOOL: "bool", STRING: "string",
}

NUMERIC_TYPES = (INT8, UINT8, INT16, UINT16, INT32, UINT32, INT64, UINT64, FLOAT, DOUBLE)

_ELEM_BYTES = {
    INT8: 1, UINT8: 1, INT16: 2, UINT16: 2, INT32: 4, UINT32: 4,
    INT64: 8, UINT64: 8, FLOAT: 4, DOUBLE: 8, BOOL: 1,
}

_PHYSICAL = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}

# CompressionType (dwio/nimble/common/Types.h:190-197)
UNCOMPRESSED = 0
ZSTD = 1
LZ4 = 3

CHUNK_HEADER_SIZE = 5  # dwio/nimble/common/ChunkHeader.h:28
FIXED_PREFIX_SIZE = 6  # dwio/nimble/encodings/common/EncodingPrefix.h:39
FBA_SLOP = 7  # dwio/nimble/common/FixedBitArray.cpp:26-31
BBP_RAW_BLOCK = 255  # BlockBitPackingEncoding.h:65: bit-width marker of a block stored as raw values
BBP_MAX_BLOCK_SIZE = 1024  # dwio/nimble/common/Constants.h:51
BBP_MAX_BLOCKS = 65535  # BlockBitPackingEncoding.h:69-70

INTEGER_TYPES = (1, 2, 3, 4, 5, 6, 7, 8)


def elem_bytes(dtype: int) -> int:
    """sizeof(physicalType) for a numeric or bool data type."""
    return _ELEM_BYTES[dtype]


def physical_dtype(dtype: int):
    """numpy unsigned dtype carrying the bit pattern of `dtype`."""
    return _PHYSICAL[_ELEM_BYTES[dtype]]


def is_numeric(dtype: int) -> bool:
    return dtype in NUMERIC_TYPES
