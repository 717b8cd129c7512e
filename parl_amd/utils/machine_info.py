"""parl/utils/machine_info.py:24-120 — the functions example scripts import (`from parl.utils import machine_info`:
examples/A2C/atari_agent.py:17).  The device here is an AMD GPU seen through torch (HIP), not nvidia-smi."""
import socket

import torch

__all__ = ['get_gpu_count', 'get_ip_address', 'is_gpu_available', 'get_free_tcp_port', 'is_port_available',
           'is_xpu_available']


def get_gpu_count():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def is_gpu_available():
    return get_gpu_count() > 0


def is_xpu_available():
    return False


def get_ip_address():
    try:
        return socket.gethostbyname(socket.gethostname())
    except OSError:
        return None


def is_port_available(port):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        return s.connect_ex(('127.0.0.1', int(port))) != 0


def get_free_tcp_port():
    with socket.socket() as s:
        s.bind(('', 0))
        return str(s.getsockname()[1])
