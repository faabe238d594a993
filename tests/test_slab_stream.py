"""The slab driver and its handle must share one stream when the communicator moves device buffers
itself (RCCL): checked here without a GPU, with stand-ins for the handle, the stream and the transport."""
import torch

from fem_amd import slabs


class FakeStream:
    cuda_stream = 0x5EED


class FakeGrid:
    NY = 8

    def __init__(self):
        self.stream, self.halo = None, None

    def set_stream(self, ptr):
        self.stream = ptr

    def set_halo_buffers(self, *a):
        self.halo = a


class DeviceComm:
    stage = False       # hands device buffers to the transport as they are


class StagedComm:
    stage = True        # copies through host memory


def _host_zeros(monkeypatch):
    real = torch.zeros
    monkeypatch.setattr(slabs.torch, "zeros", lambda n, dtype=None, device=None: real(n, dtype=dtype))


def test_device_side_comm_binds_the_handle_to_the_slab_stream(monkeypatch):
    _host_zeros(monkeypatch)
    g, st = FakeGrid(), FakeStream()
    s = slabs.SlabGrid(g, rank=0, world=2, device=torch.device("cuda", 0), comm=DeviceComm(), stream=st, ny=8, halo_rows=2)
    assert s.device_side_comm()
    assert s.stream is st and g.stream == st.cuda_stream          # the handle's kernels run on the stream the transfers are ordered on
    assert g.halo is not None and g.halo[4] == 2 * 8 * 3


def test_staged_or_host_comm_leaves_the_handle_alone(monkeypatch):
    _host_zeros(monkeypatch)
    monkeypatch.setattr(slabs.torch.cuda, "current_stream", lambda dev=None: type("S", (), {"synchronize": lambda self: None})())
    g = FakeGrid()
    s = slabs.SlabGrid(g, rank=1, world=2, device=torch.device("cuda", 0), comm=StagedComm(), ny=8, halo_rows=2)
    assert not s.device_side_comm() and s.stream is None and g.stream is None
    g2 = FakeGrid()
    s2 = slabs.SlabGrid(g2, rank=1, world=2, device=torch.device("cpu"), comm=DeviceComm(), ny=8, halo_rows=2)
    assert not s2.device_side_comm() and g2.stream is None


def test_torchcomm_stages_only_for_gloo_on_a_device(monkeypatch):
    monkeypatch.setattr(slabs.dist, "get_backend", lambda group=None: "nccl")
    assert slabs.TorchComm(0, 2, torch.device("cuda", 0)).stage is False
    monkeypatch.setattr(slabs.dist, "get_backend", lambda group=None: "gloo")
    assert slabs.TorchComm(0, 2, torch.device("cuda", 0)).stage is True
    assert slabs.TorchComm(0, 2, torch.device("cpu")).stage is False
