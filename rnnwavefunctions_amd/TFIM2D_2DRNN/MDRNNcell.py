"""``from MDRNNcell import MDRNNcell`` (2DTFIM_2DRNN/MDRNNcell.py:6-66).

The cell arithmetic elu(x_h Uh + h_h Wh + x_v Uv + h_v Wv + b) runs inside the HIP kernels
(csrc/mdrnn_kernels.h); this class only carries the constructor arguments the reference passes
(2DTFIM_2DRNN/RNNwavefunction.py:32) so that ``cell=MDRNNcell`` keeps working."""


class MDRNNcell:
    def __init__(self, num_units=None, num_in=None, name=None, dtype=None, reuse=None):
        self._num_units = num_units
        self._num_in = num_in
        self.name = name
        self.dtype = dtype

    @property
    def input_size(self):
        return self._num_in

    @property
    def state_size(self):
        return self._num_units

    @property
    def output_size(self):
        return self._num_units
