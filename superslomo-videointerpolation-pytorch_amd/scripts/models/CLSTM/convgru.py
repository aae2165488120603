"""ConvGRU / ConvBGRU with the parameter layout of the package the reference imports as
models.CLSTM.convgru (cell: conv_gates over cat[x, h] -> reset | update, conv_can over cat[x, reset*h])."""
import torch.nn as nn

from ._common import BidirectionalBottleneck


class ConvGRUCell(nn.Module):
    def __init__(self, input_dim, hidden_dim, kernel_size, bias=True):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        pad = (kernel_size[0] // 2, kernel_size[1] // 2)
        self.conv_gates = nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, kernel_size, padding=pad, bias=bias)
        self.conv_can = nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, padding=pad, bias=bias)


class ConvGRU(nn.Module):
    """Parameter container of one direction: cell_list.<l>.{conv_gates,conv_can}.{weight,bias}."""

    def __init__(self, in_channels, hidden_channels, kernel_size, num_layers, bias=True, batch_first=False):
        super().__init__()
        self.cell_list = nn.ModuleList(
            ConvGRUCell(in_channels if l == 0 else hidden_channels, hidden_channels, kernel_size, bias) for l in range(num_layers))


class ConvBGRU(BidirectionalBottleneck):
    KIND = "CGRU"
    NET = ConvGRU
