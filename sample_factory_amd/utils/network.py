"""`is_udp_port_available` (sample_factory/utils/network.py:6-15): multi-player env integrations probe a port before they
start a game server on it (sf_examples/vizdoom/doom/multiplayer/doom_multiagent.py)."""
from __future__ import annotations

import socket

from sample_factory_amd.utils.utils import log


def is_udp_port_available(port) -> bool:
    with socket.socket(socket.AF_INET, socket.SOCK_DGRAM) as sock:
        try:
            sock.bind(("", port))
        except OSError as exc:
            log.warning(f"UDP port {port} cannot be used {exc}")
            return False
    return True
