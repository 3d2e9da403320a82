"""Synthetic V3C sample streams for the container tests, built from the syntax tables the reference's writer follows (v3cUnitHeader,
PCCBitstreamWriter.cpp:309-333; sampleStreamV3CHeader / sampleStreamV3CUnit, :1492-1507) - an independent Python construction, so that the C readers
of the product and of the oracle are both checked against it. V3C_VPS and V3C_AD payloads are opaque bytes here: the transcoder carries them over."""
import struct
import numpy as np
import oracle_lib as O
import synth

VPS, AD, OVD, GVD, AVD = range(5)


def unit_header(t, psid=0, atlas=0, attr_idx=0, attr_dim=0, map_idx=0, aux=0):
    h = t << 27
    if t in (AD, OVD, GVD, AVD):
        h |= (psid << 23) | (atlas << 17)
    if t == AVD:
        h |= (attr_idx << 10) | (attr_dim << 5) | (map_idx << 1) | aux
    elif t == GVD:
        h |= (map_idx << 13) | (aux << 12)
    return struct.pack(">I", h)


def sample_stream(units, precision):
    out = bytes([(precision - 1) << 5])
    for u in units:
        out += len(u).to_bytes(precision, "big") + u
    return out


def parse(data):
    """-> (precision, [unit bytes])"""
    p = (data[0] >> 5) + 1
    pos, units = 1, []
    while pos < len(data):
        n = int.from_bytes(data[pos:pos + p], "big"); pos += p
        units.append(data[pos:pos + n]); pos += n
    assert pos == len(data)
    return p, units


def gof_streams(w, h, n_pc, seed, log2_ctb=6):
    """[occupancy, geometry, attribute] Annex-B sub-bitstreams of one GOF at R5-like settings (precision 2 occupancy, lossless)"""
    geo, attr, occ = synth.make_gof(w, h, n_pc, seed)
    return [O.encode(occ, w // 2, h // 2, 8, 8, gop=1, lossless=1, i_qp_offset=0, log2_ctb=log2_ctb, rows_per_slice=0)[0],
            O.encode(geo, w, h, 10, 16, gop=2, log2_ctb=log2_ctb, rows_per_slice=0)[0], O.encode(attr, w, h, 10, 22, gop=2, log2_ctb=log2_ctb, rows_per_slice=0)[0]]


def gof_units(streams, seed, aux=False, extra_attr_partition=False):
    """Units of one GOF in the order PCCBitstreamWriter::encode emits them (:96-237): VPS, AD, OVD, GVD (, GVD aux), AVD (, AVD aux, AVD partition 1)"""
    rng = np.random.default_rng(seed)
    blob = lambda n: bytes(rng.integers(0, 256, n, dtype=np.uint8))
    ss = [O.byte_to_sample_stream(s) for s in streams]
    u = [unit_header(VPS) + blob(int(rng.integers(20, 60))), unit_header(AD) + blob(int(rng.integers(100, 4000))),
         unit_header(OVD) + ss[0], unit_header(GVD) + ss[1]]
    if aux:
        u.append(unit_header(GVD, aux=1) + blob(300))
    u.append(unit_header(AVD) + ss[2])
    if aux:
        u.append(unit_header(AVD, aux=1) + blob(200))
    if extra_attr_partition:
        u.append(unit_header(AVD, attr_dim=1) + blob(150))
    return u
