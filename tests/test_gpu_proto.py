"""mapping::proto::HybridGrid wire format of device grids against google.protobuf's own
serialisation of the same message (descriptor built from mapping/proto/3d/hybrid_grid.proto) with
the cells in the reference iterator's order (the oracle's ForEachCell)."""
import numpy as np
import pytest

from helpers import build_oracle_submap, oracle_cells_dict, to_device_grid

pytestmark = pytest.mark.gpu


def hybrid_grid_message_class():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto()
    f.name = "cartographer/mapping/proto/3d/hybrid_grid.proto"
    f.package = "cartographer.mapping.proto"
    f.syntax = "proto3"
    m = f.message_type.add()
    m.name = "HybridGrid"
    T = descriptor_pb2.FieldDescriptorProto
    for name, number, typ, label in (("resolution", 1, T.TYPE_FLOAT, T.LABEL_OPTIONAL),
                                     ("x_indices", 3, T.TYPE_SINT32, T.LABEL_REPEATED),
                                     ("y_indices", 4, T.TYPE_SINT32, T.LABEL_REPEATED),
                                     ("z_indices", 5, T.TYPE_SINT32, T.LABEL_REPEATED),
                                     ("values", 6, T.TYPE_INT32, T.LABEL_REPEATED)):
        fd = m.field.add()
        fd.name, fd.number, fd.type, fd.label = name, number, typ, label
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("cartographer.mapping.proto.HybridGrid"))


@pytest.fixture(scope="module")
def dl():
    import dliom
    return dliom


@pytest.fixture(scope="module")
def ctx(dl):
    c = dl.Context(0)
    yield c
    c.close()


def reference_bytes(orc_grid):
    msg = hybrid_grid_message_class()()
    msg.resolution = orc_grid.resolution
    xyz, v = orc_grid.export_cells()  # the reference iterator's order
    msg.x_indices.extend(int(c) for c in xyz[:, 0])
    msg.y_indices.extend(int(c) for c in xyz[:, 1])
    msg.z_indices.extend(int(c) for c in xyz[:, 2])
    msg.values.extend(int(x) for x in v)
    return msg.SerializeToString()


def test_to_proto_is_byte_identical_and_round_trips(dl, ctx, orc):
    og = build_oracle_submap(orc, 0.1, num_scans=3, beams=16, azimuths=128, max_range=20.0)
    # negative, large and meta-cell-crossing indices; a grid that has grown (bits > 1)
    og.set_values(np.array([[-700, 3, 64], [700, -1, -65], [63, 64, -64], [0, 0, 0]]), np.array([1, 32767, 12345, 77], np.uint16))
    dg = to_device_grid(dl, ctx, og)
    want = reference_bytes(og)
    got = dg.to_proto()
    assert got == want
    back = dl.HybridGrid.from_proto(ctx, got)
    assert abs(back.resolution - og.resolution) == 0
    assert back.cells() == oracle_cells_dict(og)
    # what google.protobuf parses out of our bytes is the message itself
    msg = hybrid_grid_message_class()()
    msg.ParseFromString(got)
    assert len(msg.values) == len(oracle_cells_dict(og))
    back.close()
    dg.close()


def test_from_proto_accepts_unpacked_fields_and_empty_grids(dl, ctx, orc):
    empty = dl.HybridGrid(ctx, 0.25)
    data = empty.to_proto()
    assert data == bytes([0x0D]) + np.float32(0.25).tobytes()
    g = dl.HybridGrid.from_proto(ctx, data)
    assert g.cells() == {} and g.resolution == 0.25
    g.close()
    empty.close()
    # unpacked encoding of the same repeated fields (tag per element) + an unknown field
    def varint(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80)
            v >>= 7
        out.append(v)
        return bytes(out)
    zz = lambda n: (n << 1) ^ (n >> 31)
    raw = bytes([0x0D]) + np.float32(0.5).tobytes()
    for (x, y, z, v) in ((1, -2, 3, 20000), (-9, 8, -7, 3)):
        raw += bytes([3 << 3]) + varint(zz(x) & 0xFFFFFFFF) + bytes([4 << 3]) + varint(zz(y) & 0xFFFFFFFF)
        raw += bytes([5 << 3]) + varint(zz(z) & 0xFFFFFFFF) + bytes([6 << 3]) + varint(v)
    raw += bytes([(9 << 3) | 0]) + varint(5)
    g = dl.HybridGrid.from_proto(ctx, raw)
    assert g.cells() == {(1, -2, 3): 20000, (-9, 8, -7): 3}
    g.close()
    with pytest.raises(dl.DliomError):  # CHECK_EQ(values_size, x_indices_size)
        dl.HybridGrid.from_proto(ctx, bytes([0x0D]) + np.float32(0.5).tobytes() + bytes([6 << 3]) + varint(7))
