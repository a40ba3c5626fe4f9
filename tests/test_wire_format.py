"""The socket wire format against byte-level vectors captured by driving the reference's own
handle_client (socket_server_para.py:137-195) through a fake connection (tools/gen_golden.py, GW).  CPU only."""
import json

import numpy as np
import pytest

from conftest import load_golden
from dipole_normal_prop_amd import wire


@pytest.fixture(scope="module")
def gw():
    return load_golden("GW_wire_format")


def test_constants_match_the_reference_bytes(gw):
    assert wire.ACK == gw["ack"].tobytes() == gw["unknown_ack"].tobytes() == gw["short_ack"].tobytes()
    assert wire.ERROR == gw["unknown_reply"].tobytes() == gw["short_reply"].tobytes()
    assert wire.REQUEST_BUFFER_SIZE == 1000 and wire.BYTES_PER_POINT == 24


def test_request_encode_decode(gw):
    header, payload = wire.encode_request("simple_estimate", {"diffuse": True}, gw["xyz"])
    assert header == gw["header"].tobytes()                 # json.dumps of the same dict, key order included
    assert payload == gw["payload"].tobytes()
    req = wire.decode_header(gw["header"].tobytes())
    assert req == {"function_name": "simple_estimate", "function_config": {"diffuse": True}, "data_size": 7}
    assert wire.payload_size(req) == 7 * 24 == len(payload)
    xyz = wire.decode_payload(req, payload)
    assert xyz.dtype == np.float64 and np.array_equal(xyz, gw["decoded_xyz"])
    header32, payload32 = wire.encode_request("simple_estimate", {"diffuse": True}, gw["xyz"].astype(np.float32))
    assert len(payload32) == 7 * 24                         # always float64 on the wire


def test_reply_encode_decode_and_serve(gw):
    assert wire.encode_reply(gw["result"]) == gw["reply"].tobytes()         # float32 result -> float64 bytes
    out = wire.decode_reply(gw["reply"].tobytes(), 7)
    assert out.shape == (7, 6) and np.array_equal(out, gw["result"].astype(np.float64))
    seen = {}

    def handler(xyz, config):
        seen["xyz"], seen["config"] = xyz, config
        return gw["result"]

    reply = wire.serve_request(gw["header"].tobytes(), gw["payload"].tobytes(), {"simple_estimate": handler})
    assert reply == gw["reply"].tobytes() and int(gw["n_sent_ok"]) == 2     # ack + reply, nothing else
    assert np.array_equal(seen["xyz"], gw["xyz"]) and seen["config"] == {"diffuse": True}


def test_failures_are_the_error_reply(gw, capsys):
    payload = gw["payload"].tobytes()
    bad = json.dumps({"function_name": "nope", "function_config": {}, "data_size": 7}).encode()
    assert wire.serve_request(bad, payload, {"simple_estimate": lambda x, c: x}) == wire.ERROR
    assert wire.serve_request(gw["header"].tobytes(), payload[:100], {"simple_estimate": lambda x, c: x}) == wire.ERROR
    assert "Data size mismatch. Expected 168 bytes, but received 100 bytes." in capsys.readouterr().out
    assert wire.serve_request(b"not json", payload, {}) == wire.ERROR

    def boom(xyz, config):
        raise RuntimeError("estimator failed")

    assert wire.serve_request(gw["header"].tobytes(), payload, {"simple_estimate": boom}) == wire.ERROR
    assert wire.serve_request(gw["header"].tobytes(), b"", {"simple_estimate": boom}) == b""   # closed, no reply
    with pytest.raises(wire.WireError):
        wire.decode_reply(wire.ERROR, 7)
    with pytest.raises(wire.WireError):
        wire.decode_reply(b"\0" * 47, 1)
    with pytest.raises(wire.WireError):
        wire.encode_request("f", {"pad": "x" * 1000}, np.zeros((1, 3)))      # header must fit the 1000-byte read
    # the N^2 admission budget of apply_pts
    n = 30001
    big_header, _ = wire.encode_request("simple_estimate", {"diffuse": True}, np.zeros((1, 3)))
    req = json.loads(big_header)
    req["data_size"] = n
    assert wire.serve_request(json.dumps(req).encode(), b"\0" * (n * 24), {"simple_estimate": boom}) == wire.ERROR
