import ctypes as C
l = C.CDLL("mundy_amd/lib/libmundy_hip.so")
n = C.c_int(); name = C.create_string_buffer(64)
rc = l.mhip_device_info(C.byref(n), name, 64)
l.mhip_last_error.restype = C.c_char_p
print("no-torch device_info rc", rc, n.value, name.value, l.mhip_last_error())
