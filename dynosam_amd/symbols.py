"""64-bit variable keys, bit-exact with the reference.

Mirrors dynosam_opt/include/dynosam_opt/Symbols.hpp:14-20,126-151 and
dynosam_opt/src/Symbols.cc:160-175 (Cantor pairing), on top of the GTSAM-4.2.0
``Symbol`` / ``LabeledSymbol`` bit layout (chr in bits 56-63, label in bits 48-55).
"""
from __future__ import annotations

import math

kPoseSymbolChar = ord("X")
kVelocitySymbolChar = ord("V")
kObjectMotionSymbolChar = ord("H")
kObjectPoseSymbolChar = ord("L")
kStaticLandmarkSymbolChar = ord("l")
kDynamicLandmarkSymbolChar = ord("m")
kImuBiasSymbolChar = ord("b")

_IDX56 = (1 << 56) - 1
_IDX48 = (1 << 48) - 1


def symbol(c: int, j: int) -> int:
    """gtsam::Symbol(c, j).key()"""
    if j < 0 or j > _IDX56:
        raise ValueError("Symbol index does not fit in 56 bits")
    return (c << 56) | j


def labeled_symbol(c: int, label: int, j: int) -> int:
    """gtsam::LabeledSymbol(c, label, j).key()"""
    if j < 0 or j > _IDX48:
        raise ValueError("LabeledSymbol index does not fit in 48 bits")
    return (c << 56) | ((label & 0xFF) << 48) | j


def symbol_chr(key: int) -> int:
    return (key >> 56) & 0xFF


def symbol_index(key: int) -> int:
    return key & _IDX56


def labeled_label(key: int) -> int:
    return (key >> 48) & 0xFF


def labeled_index(key: int) -> int:
    return key & _IDX48


def cantor_pair(k1: int, k2: int) -> int:
    """CantorPairingFunction::pair (Symbols.cc:160-164)."""
    return ((k1 + k2) * (k1 + k2 + 1) // 2) + k2


def cantor_depair(z: int) -> tuple[int, int]:
    """CantorPairingFunction::depair (Symbols.cc:166-175): double-precision sqrt, as the
    reference (exact while 8z+1 < 2**53)."""
    w = int(math.floor((math.sqrt(float(z * 8 + 1)) - 1) / 2))
    t = (w * (w + 1)) // 2
    k2 = z - t
    k1 = w - k2
    return k1, k2


def CameraPoseSymbol(frame_id: int) -> int:
    return symbol(kPoseSymbolChar, frame_id)


def StaticLandmarkSymbol(tracklet_id: int) -> int:
    return symbol(kStaticLandmarkSymbolChar, tracklet_id)


def DynamicLandmarkSymbol(frame_id: int, tracklet_id: int) -> int:
    """DynamicPointSymbol('m', tracklet, frame) — Symbols.hpp:133-136."""
    if tracklet_id == -1:
        raise ValueError("DynamicPointSymbol cannot be constructed from invalid tracklet id (-1)")
    j = cantor_pair(tracklet_id, frame_id)
    if cantor_depair(j) != (tracklet_id, frame_id):  # the reference CHECKs this round trip
        raise ValueError("Cantor depair round trip failed")
    return symbol(kDynamicLandmarkSymbolChar, j)


def HybridDynamicKey(tracklet_id: int) -> int:
    """HybridFormulationProperties::makeDynamicKey (HybridEstimator.hpp:1154-1158)."""
    return DynamicLandmarkSymbol(0, tracklet_id)


def ObjectMotionSymbol(object_id: int, frame_id: int) -> int:
    """Symbols.hpp:140-143: label = object_id + '0' as unsigned char."""
    return labeled_symbol(kObjectMotionSymbolChar, (object_id + ord("0")) & 0xFF, frame_id)


def ObjectPoseSymbol(object_id: int, frame_id: int) -> int:
    return labeled_symbol(kObjectPoseSymbolChar, (object_id + ord("0")) & 0xFF, frame_id)


def reconstructMotionInfo(key: int):
    """reconstructMotionInfo (Symbols.hpp:33-34): (ok, object_id, frame_id)."""
    if symbol_chr(key) != kObjectMotionSymbolChar:
        return False, None, None
    return True, labeled_label(key) - ord("0"), labeled_index(key)


def reconstructPoseInfo(key: int):
    if symbol_chr(key) != kObjectPoseSymbolChar:
        return False, None, None
    return True, labeled_label(key) - ord("0"), labeled_index(key)


def DynoChrExtractor(key: int) -> int:
    """Symbols.hpp:30 — the symbol character for any DynoSAM key, 0 if none."""
    c = symbol_chr(key)
    return c if c in (kPoseSymbolChar, kVelocitySymbolChar, kObjectMotionSymbolChar, kObjectPoseSymbolChar,
                      kStaticLandmarkSymbolChar, kDynamicLandmarkSymbolChar, kImuBiasSymbolChar) else 0
