"""ctypes mirror of include/dellyhip.h (the C-ABI of the MI355X split-read
refinement path).  Pure declarations: the structs are shared by the product
bindings (delly_amd.refine) and by the test-only checker bindings under oracle/,
so parity tests compare identical record layouts.

Reference types mirrored: the duck-typed TConfig fields (SURVEY.md 8b,
src/delly.h:49-82), torali::StructuralVariantRecord (src/tags.h:93-130) and
torali::AlignDescriptor (src/split.h:15-25).
"""
import ctypes as C

OK = 0
E_NODEVICE = -1
E_ARG = -2
E_RUNTIME = -3
E_LIMIT = -4
E_NOMEM = -5


class Params(C.Structure):
    _fields_ = [
        ("match", C.c_int32),
        ("mismatch", C.c_int32),
        ("gap_open", C.c_int32),
        ("gap_extend", C.c_int32),
        ("min_clique_size", C.c_int32),
        ("minimum_flank_size", C.c_int32),
        ("indelsize", C.c_int32),
        ("min_cons_window", C.c_int32),
        ("flank_quality", C.c_float),
        ("reserved", C.c_int32),
    ]


REALIGN = 1           # dellyhip_params.reserved bit 0 (DELLYHIP_REALIGN)
COMPACT_ALLELES = 4   # bit 2 (DELLYHIP_COMPACT_ALLELES): no "REF,ALT" bytes where they are plain substrings; allele_len = -(length)


def params_sr(compact_alleles=False):
    """`delly sr` defaults: src/delly.h:221,240,393-398."""
    return Params(5, -4, -10, -1, 2, 13, 1000, 100, 0.95, COMPACT_ALLELES if compact_alleles else 0)


def params_lr(realign=False):
    """`delly lr` defaults: src/tegua.h:230-241 (aliscore unused on that path).
    realign: the `realign` argument of alignConsensus (src/split.h:644-646, true on the
    long-read call sites src/assemble.h:849,916) travels as bit 0 of `reserved`."""
    return Params(5, -4, -10, -1, 3, 100, 10000, 1000, 0.9, 1 if realign else 0)


class Junction(C.Structure):
    _fields_ = [
        ("svid", C.c_int32),
        ("svt", C.c_int32),
        ("chr", C.c_int32),
        ("chr2", C.c_int32),
        ("sv_start", C.c_int32),
        ("sv_end", C.c_int32),
        ("ins_len", C.c_int32),
        ("n_seq", C.c_int32),
        ("seq_first", C.c_uint64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("svid", C.c_int32),
        ("ok", C.c_int32),
        ("sv_start", C.c_int32),
        ("sv_end", C.c_int32),
        ("ci_wiggle", C.c_int32),
        ("ins_len", C.c_int32),
        ("cons_bp", C.c_int32),
        ("hom_len", C.c_int32),
        ("sr_support", C.c_int32),
        ("sr_align_quality", C.c_float),
        ("matches", C.c_int32),
        ("mismatches", C.c_int32),
        ("c_start", C.c_int32),
        ("c_end", C.c_int32),
        ("r_start", C.c_int32),
        ("r_end", C.c_int32),
        ("hom_left", C.c_int32),
        ("hom_right", C.c_int32),
        ("score_unsplit", C.c_int32),
        ("score_best", C.c_int32),
        ("cons_left", C.c_int32),
        ("ref_left", C.c_int32),
        ("ref_right", C.c_int32),
        ("cons_len", C.c_int32),
        ("ref_len", C.c_int32),
        ("cons_off", C.c_uint64),
        ("allele_off", C.c_uint64),
        ("aln_off", C.c_uint64),
        ("allele_len", C.c_int32),
        ("aln_len", C.c_int32),
        ("status", C.c_int32),
        ("reserved", C.c_int32),
    ]


class AlignJob(C.Structure):
    """dellyhip_align_job: AlignJob of src/coverage.h:87-96 with the strings as blob ranges."""
    _fields_ = [
        ("cons_off", C.c_uint64),
        ("ref_off", C.c_uint64),
        ("seq_off", C.c_uint64),
        ("cons_len", C.c_uint32),
        ("ref_len", C.c_uint32),
        ("seq_len", C.c_uint32),
        ("file_index", C.c_uint32),
        ("sv_id", C.c_uint32),
        ("qual", C.c_uint8),
        ("pad0", C.c_uint8),
        ("pad1", C.c_uint8),
        ("pad2", C.c_uint8),
    ]


class AlignResult(C.Structure):
    """dellyhip_align_result: AlignResult of src/coverage.h:98-105 plus the two edlib distances."""
    _fields_ = [
        ("file_index", C.c_uint32),
        ("sv_id", C.c_uint32),
        ("dist_alt", C.c_int32),
        ("dist_ref", C.c_int32),
        ("type", C.c_uint8),
        ("qual", C.c_uint8),
        ("status", C.c_int16),
    ]


class NwJob(C.Structure):
    """dellyhip_nw_job: one _editDistanceNW(query, target) call (src/genotype.h:21-30)."""
    _fields_ = [
        ("query_off", C.c_uint64),
        ("target_off", C.c_uint64),
        ("query_len", C.c_uint32),
        ("target_len", C.c_uint32),
    ]


class Probes(C.Structure):
    """dellyhip_probes: consProbeArr / refProbeArr / BpRegion of _generateProbes (src/coverage.h:230-258)."""
    _fields_ = [
        ("svid", C.c_int32),
        ("ok", C.c_int32),
        ("hom_left", C.c_int32),
        ("hom_right", C.c_int32),
        ("region_start0", C.c_int32), ("region_start1", C.c_int32),
        ("region_end0", C.c_int32), ("region_end1", C.c_int32),
        ("bppos0", C.c_int32), ("bppos1", C.c_int32),
        ("cons_len0", C.c_int32), ("cons_len1", C.c_int32),
        ("ref_len0", C.c_int32), ("ref_len1", C.c_int32),
        ("cons_off0", C.c_uint64), ("cons_off1", C.c_uint64),
        ("ref_off0", C.c_uint64), ("ref_off1", C.c_uint64),
        ("status", C.c_int32),
        ("reserved", C.c_int32),
    ]


# numpy structured dtypes with the same layout (for vectorised comparisons)
def _np_dtype(struct):
    import numpy as np

    m = {C.c_int32: "<i4", C.c_uint64: "<u8", C.c_float: "<f4", C.c_uint32: "<u4", C.c_uint8: "u1", C.c_int16: "<i2"}
    names, formats, offsets = [], [], []
    for name, typ in struct._fields_:
        names.append(name)
        formats.append(m[typ])
        offsets.append(getattr(struct, name).offset)
    return np.dtype({"names": names, "formats": formats, "offsets": offsets,
                     "itemsize": C.sizeof(struct)})


def junction_dtype():
    return _np_dtype(Junction)


def result_dtype():
    return _np_dtype(Result)


def align_job_dtype():
    return _np_dtype(AlignJob)


def align_result_dtype():
    return _np_dtype(AlignResult)


def nw_job_dtype():
    return _np_dtype(NwJob)


def probes_dtype():
    return _np_dtype(Probes)
