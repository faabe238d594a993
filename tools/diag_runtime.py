import sys, os, ctypes
sys.path.insert(0, '/root/repo')
mode = sys.argv[1]
def maps():
    s=set()
    for l in open('/proc/self/maps'):
        if 'amdhip' in l or 'hsa-runtime' in l: s.add(l.split()[-1])
    return sorted(s)
try:
    if mode == 'torch_first':
        import torch; print('avail', torch.cuda.is_available()); torch.cuda.set_device(0)
        x = torch.zeros(4, device='cuda'); print('torch alloc ok')
        import fem_amd; g = fem_amd.ColumnGrid(64,64,2,2); print('grid ok', g.NX)
    elif mode == 'lib_first':
        import fem_amd; fem_amd.load_library(); print('lib loaded')
        import torch; print('avail', torch.cuda.is_available()); torch.cuda.set_device(0)
        g = fem_amd.ColumnGrid(64,64,2,2); print('grid ok', g.NX)
        x = torch.zeros(4, device='cuda'); print('torch alloc ok')
    elif mode == 'lib_first_grid_first':
        import fem_amd; g = fem_amd.ColumnGrid(64,64,2,2); print('grid ok', g.NX)
        import torch; print('avail', torch.cuda.is_available()); torch.cuda.set_device(0)
        x = torch.zeros(4, device='cuda'); print('torch alloc ok')
except Exception as e:
    print('FAIL', mode, repr(e)[:300])
print(mode, maps())
