"""Wire format of the reference's socket service (socket_server_para.py:137-195, socket_server.py) as pure
encode / decode functions - no sockets, no threads: the TCP server itself is out of scope, the byte layout of
what travels is part of the path's data formats.

One request, in the order the bytes cross the connection:

    client -> server   header   UTF-8 JSON {"function_name": str, "function_config": {...}, "data_size": N},
                                read with ONE recv of REQUEST_BUFFER_SIZE = 1000 bytes, so it must fit in 1000
    server -> client   ack      the JSON {"status": "OK"}
    client -> server   payload  N * 24 bytes: N rows of three little-endian float64 (x, y, z), C order
    server -> client   reply    N * 48 bytes: N rows of six float64 (x, y, z, nx, ny, nz)   - or, on any
                                failure, the JSON {"status": "ERROR"}

`serve_request` strings these together around a table of handlers the way handle_client does (unknown function
names and exceptions become the ERROR reply; the admission budget `apply_pts` of the reference, N^2 <= 30000^2,
is enforced as well).
"""
import json

import numpy as np

REQUEST_BUFFER_SIZE = 1000                 # socket_server_para.py:13
BYTES_PER_POINT = 24                       # three float64
MAX_POINT_PAIRS = 30000 * 30000            # socket_server_para.py:16 (max_pts): N*N above this is refused
ACK = json.dumps({"status": "OK"}).encode()
ERROR = json.dumps({"status": "ERROR"}).encode()


class WireError(ValueError):
    pass


def encode_request(function_name: str, function_config: dict, xyz) -> tuple:
    """(header bytes, payload bytes) for a cloud xyz[N,3] (any float dtype; sent as float64)."""
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64))
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise WireError(f"xyz must be [N,3], got {xyz.shape}")
    header = json.dumps({"function_name": function_name, "function_config": function_config,
                         "data_size": int(xyz.shape[0])}).encode()
    if len(header) > REQUEST_BUFFER_SIZE:
        raise WireError(f"header of {len(header)} bytes exceeds the server's {REQUEST_BUFFER_SIZE}-byte read")
    return header, xyz.tobytes()


def decode_header(raw: bytes) -> dict:
    """The request dict from the first (<= 1000-byte) read.  Raises WireError when it is not the expected JSON."""
    try:
        req = json.loads(raw[:REQUEST_BUFFER_SIZE].decode())
        int(req["data_size"]), req["function_name"], req["function_config"]
    except Exception as exc:
        raise WireError(f"malformed request header: {exc}") from exc
    return req


def payload_size(req: dict) -> int:
    return int(req["data_size"]) * BYTES_PER_POINT


def decode_payload(req: dict, data: bytes) -> np.ndarray:
    """xyz[N,3] float64 from the payload bytes; the length must be exactly data_size * 24 (the reference asserts)."""
    if len(data) != payload_size(req):
        raise WireError(f"Data size mismatch. Expected {payload_size(req)} bytes, but received {len(data)} bytes.")
    return np.frombuffer(data, dtype=np.float64).reshape(-1, 3)


def encode_reply(result) -> bytes:
    """The reply for an oriented cloud [N,6] (any float dtype): float64, C order."""
    return np.asarray(result).astype(np.float64).tobytes()


def decode_reply(raw: bytes, n_points: int) -> np.ndarray:
    """Oriented cloud [N,6] float64 from a reply; raises WireError on the ERROR reply or a wrong length."""
    if raw == ERROR:
        raise WireError("server replied ERROR")
    if len(raw) != n_points * 48:
        raise WireError(f"reply of {len(raw)} bytes, expected {n_points * 48}")
    return np.frombuffer(raw, dtype=np.float64).reshape(n_points, 6)


def serve_request(header: bytes, payload: bytes, handlers: dict) -> bytes:
    """What handle_client sends back after the ack for one request: handlers maps function_name ->
    f(xyz[N,3] float64, function_config) -> [N,6]; every failure (malformed header, size mismatch, budget,
    unknown function, handler exception) is the ERROR reply, as in the reference's try/except."""
    try:
        req = decode_header(header)
        if not payload:                       # `if not data: return` - the reference closes without a reply
            return b""
        xyz = decode_payload(req, payload)
        if len(xyz) * len(xyz) > MAX_POINT_PAIRS:
            raise WireError(f"Too many points to process at once. {len(xyz) ** 2} points requested, "
                            f"but only {MAX_POINT_PAIRS} points allowed.")
        fn = handlers.get(req["function_name"])
        if fn is None:
            raise WireError(f"Unknown method: {req['function_name']}")
        return encode_reply(fn(xyz, req["function_config"]))
    except Exception as exc:
        print(f"Error: {exc}")
        return ERROR
