"""Minimal ctypes access to the HIP runtime (used only by the host-staged test transport)."""
import ctypes as C

_hip = None


def _lib():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise ImportError("libamdhip64.so not found")
    return _hip


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with HIP error {rc}")


def stream_synchronize(stream):
    _chk(_lib().hipStreamSynchronize(C.c_void_p(stream)), "hipStreamSynchronize")


def memcpy_dtoh(dst_host, src_dev, nbytes):
    _chk(_lib().hipMemcpy(C.c_void_p(dst_host), C.c_void_p(src_dev), C.c_size_t(nbytes), C.c_int(2)), "hipMemcpy D2H")


def memcpy_htod(dst_dev, src_host, nbytes):
    _chk(_lib().hipMemcpy(C.c_void_p(dst_dev), C.c_void_p(src_host), C.c_size_t(nbytes), C.c_int(1)), "hipMemcpy H2D")


def current_device_pci_bus_id():
    """'0000:c1:00.0' of the HIP device this thread is bound to (sysfs: /sys/bus/pci/devices/<id>)."""
    dev = C.c_int(0)
    _chk(_lib().hipGetDevice(C.byref(dev)), "hipGetDevice")
    buf = C.create_string_buffer(64)
    _chk(_lib().hipDeviceGetPCIBusId(buf, C.c_int(64), dev), "hipDeviceGetPCIBusId")
    return buf.value.decode().lower()
