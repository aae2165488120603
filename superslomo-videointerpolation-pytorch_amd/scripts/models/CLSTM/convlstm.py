"""ConvLSTM / ConvBLSTM with the parameter layout of the package the reference imports as
models.CLSTM.convlstm (cell: one Conv2d over cat[x, h] -> 4*hidden channels, gate order i, f, o, g)."""
import torch.nn as nn

from ._common import BidirectionalBottleneck


class ConvLSTMCell(nn.Module):
    def __init__(self, input_dim, hidden_dim, kernel_size, bias=True):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        pad = (kernel_size[0] // 2, kernel_size[1] // 2)
        self.conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size, padding=pad, bias=bias)


class ConvLSTM(nn.Module):
    """Parameter container of one direction: cell_list.<l>.conv.{weight,bias}."""

    def __init__(self, in_channels, hidden_channels, kernel_size, num_layers, bias=True, batch_first=False):
        super().__init__()
        self.cell_list = nn.ModuleList(
            ConvLSTMCell(in_channels if l == 0 else hidden_channels, hidden_channels, kernel_size, bias) for l in range(num_layers))


class ConvBLSTM(BidirectionalBottleneck):
    KIND = "CLSTM"
    NET = ConvLSTM
