"""The `.mcq` container on the GPU box: the byte-for-byte pins of tests/test_mcq_container.py run in the
driver's `-m gpu` set too, plus the container around a REAL compress of a 768x512 image on the device: header bytes as
the schema dictates, `File.serialize -> deserialize -> decompress` restores the same pixels as `decode(codes)`."""
import pytest
import torch

import test_mcq_container as T
from mcquic_amd.utils.specification import File

pytestmark = pytest.mark.gpu


def test_serialize_byte_for_byte():
    T.test_serialize_writes_the_schema_document_byte_for_byte()
    T.test_expected_document_is_what_the_msgpack_library_packs()


def test_deserialize_and_bin16():
    T.test_deserialize_reads_the_hand_written_document()
    T.test_larger_streams_use_bin16()
    T.test_empty_or_non_bytes_contents_are_invalid()


def test_container_around_a_device_compress(dev):
    from mcquic_amd import Compressor
    from oracle import mcquic_ref as R
    ks = [8192, 2048, 512]
    sd = R.make_state_dict(128, 2, ks, seed=0)
    model = Compressor(128, 2, ks).eval()
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    model.QuantizationParameter = "2"
    x = R.make_images(1, 768, 512).to(dev)
    codes, binaries, headers = model.compress(x)
    data = File(headers[0], binaries[0]).serialize()
    # header part of the document: everything up to the `contents` key is fixed by the schema for this geometry
    head = T.EXPECTED[:T.EXPECTED.index(T._s("contents"))]
    assert data.startswith(head)
    assert data[len(head):len(head) + 10] == T._s("contents") + b"\x93"         # three streams, one per level
    back = File.deserialize(data)
    assert back.Content == binaries[0] and back.FileHeader.CodeSize == headers[0].CodeSize
    restored = model.decompress([back.Content], [back.FileHeader])
    assert torch.equal(restored, model.decode(codes))
    # a header whose code size does not belong to its image size is refused before anything is decoded
    bad = File.deserialize(data)
    bad.FileHeader.CodeSize.heights[0] = 1 << 15
    with pytest.raises(RuntimeError):
        model.decompress([bad.Content], [bad.FileHeader])
