"""Members of <prefix>.update_DS.zip as the batched from_msa driver writes them (reference subcommands/from_msa.py:114-127
pickles every locus's PrgBuilder after build_prg(); prg_builder.py:121-166 stores the pickles in a zip).

The reference's member is a pickled object graph (Biopython records inside).  Ours never left this package either (update reads
the update_DS this package wrote), so the batched driver writes a member as what the device already holds: a small header
followed by slices of the batch's arrays — the locus's alignment matrix, its titles, its tree as 8 int32 per node in preorder
(mprg_forest_export_*), the row lists of cluster children, the PRG index — no per-node Python object is built or pickled on
the way out (30 000 loci have ~5 million nodes).  PrgBuilder.deserialize_from_bytes() materialises the reference-shaped
objects from it on the way in; members pickled the old way (update writes whole builders) are still read.
"""
import json
import struct
from typing import List

import numpy as np

MAGIC = b"MPRGDS1\n"
KIND_LEAF, KIND_INTERVAL, KIND_CLUSTER = 0, 1, 2


# the alignment of a member: "ascii" (S x C bytes, the default) or "nib4": the cell codes at four bits per cell as the device packs them
# (mprg_export_alignments: rows of ceil(C / 2) bytes, cell 2 j in the low nibble of byte j) — half the bytes of the matrix that is
# 96 % of a member (update_DS.zip of a 30 000-gene pan-genome: 6.5 -> 3.4 GB)
_CODE_ASCII = np.frombuffer(b"ACGT-RYKMSWN----", np.uint8)


def member_header(locus: str, alignment_format: str, max_nesting: int, min_match_length: int, next_node_id: int, site_num: int,
                  rows: int, cols: int, title_bytes: int, n_nodes: int, n_rows: int, n_index: int, enc: str = None) -> bytes:
    # (written by hand: a json.dumps per locus is a measurable share of a 30 000-locus run; names go through json for escaping)
    head = (f'{{"locus":{json.dumps(locus)},"format":{json.dumps(alignment_format)},"N":{max_nesting},"L":{min_match_length},'
            f'"next_node_id":{next_node_id},"site_num":{site_num},"S":{rows},"C":{cols},"titles":{title_bytes},"nodes":{n_nodes},'
            f'"rows":{n_rows},"index":{n_index}' + (f',"enc":{json.dumps(enc)}' if enc else "") + "}").encode()
    return MAGIC + struct.pack("<I", len(head)) + head


def pack_member(locus, alignment_format, max_nesting, min_match_length, next_node_id, site_num, data: np.ndarray, titles: bytes,
                records: np.ndarray, rows: np.ndarray, index: np.ndarray) -> List[memoryview]:
    """Header + segments of one member (the caller writes them back to back; nothing is concatenated here)."""
    data = np.ascontiguousarray(data)
    head = member_header(locus, alignment_format, max_nesting, min_match_length, next_node_id, site_num, data.shape[0],
                         data.shape[1], len(titles), len(records), len(rows), len(index))
    return [memoryview(head), memoryview(data).cast("B"), memoryview(titles),
            memoryview(np.ascontiguousarray(records, np.int32)).cast("B"), memoryview(np.ascontiguousarray(rows, np.int32)).cast("B"),
            memoryview(np.ascontiguousarray(index, np.int32)).cast("B")]


def is_packed(blob) -> bool:
    return bytes(blob[:len(MAGIC)]) == MAGIC


def unpack_member(blob):
    """Packed member -> PrgBuilder with the reference-shaped node objects (ids, nesting levels, PRG index, site counter)."""
    from .msa import MSA
    from .prg_builder import PrgBuilder
    from .recursion_tree import LeafNode, MultiClusterNode, MultiIntervalNode, SubAlignment
    mv = memoryview(blob)
    (hl,) = struct.unpack("<I", mv[len(MAGIC):len(MAGIC) + 4])
    pos = len(MAGIC) + 4
    h = json.loads(bytes(mv[pos:pos + hl]))
    pos += hl

    def take(nbytes):
        nonlocal pos
        out = mv[pos:pos + nbytes]
        pos += nbytes
        return out

    S, C = h["S"], h["C"]
    if h.get("enc") == "nib4":
        nb = (C + 1) // 2
        packed = np.frombuffer(take(S * nb), np.uint8).reshape(S, nb)
        cells = np.empty((S, 2 * nb), np.uint8)
        cells[:, 0::2], cells[:, 1::2] = packed & 15, packed >> 4
        data = _CODE_ASCII[cells[:, :C]]
    elif h.get("enc") in (None, "ascii"):
        data = np.frombuffer(take(S * C), np.uint8).reshape(S, C).copy()
    else:
        raise ValueError(f"update_DS member of locus {h['locus']}: unknown alignment encoding {h['enc']!r}")
    titles = bytes(take(h["titles"])).decode("ascii").split("\n")[:S]
    recs = np.frombuffer(take(32 * h["nodes"]), np.int32).reshape(-1, 8)
    rows_all = np.frombuffer(take(4 * h["rows"]), np.int32)
    index = np.frombuffer(take(12 * h["index"]), np.int32).reshape(-1, 3)
    ids = []
    for title in titles:
        words = title.split(None, 1)
        ids.append(words[0] if words else "")
    base = MSA(_data=data, _ids=ids, _descs=titles)
    nodes = [None] * len(recs)
    node_rows = [None] * len(recs)

    def root_factory(builder):
        for i, (parent, kind, level, nrows, row_at, col0, ncols, _alleles) in enumerate(recs.tolist()):
            par = nodes[parent] if parent >= 0 else None
            if nrows == -1:
                rows = None
            elif nrows == -2:
                rows = node_rows[parent]
            else:
                rows = rows_all[row_at:row_at + nrows].astype(np.int64)
            node_rows[i] = rows
            stored = SubAlignment(base, rows, col0, ncols)
            if kind == KIND_LEAF:
                node = LeafNode(level, stored, par, builder, node_id=i)
            else:
                node = (MultiIntervalNode if kind == KIND_INTERVAL else MultiClusterNode)(level, stored, par, builder, [], node_id=i)
            nodes[i] = node
            if par is not None:
                par._children.append(node)          # preorder: a node's children arrive in their order
        return nodes[0]

    builder = PrgBuilder(h["locus"], None, h["format"], h["N"], h["L"], _root_factory=root_factory)
    builder.next_node_id = h["next_node_id"]
    builder.site_num = h["site_num"]
    for a, e, nid in index.tolist():
        builder.update_PRG_index(a, e, nodes[nid])
    return builder
