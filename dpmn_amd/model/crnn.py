"""Mirror of the reference's CRNN recogniser (model/crnn/crnn.py:1-79) for ONE job on the SR path: under --arch tatt the trainer
derives TATT's `label_vecs` from the frozen CRNN's logits on the LR image (interfaces/super_resolution.py:165-169, 92-96;
parse_crnn_data base.py:419-425).  It is an input generator of the frozen PSN, not part of the refined hot path, and is built from
stock torch operators (MIOpen convolutions / LSTM on the GPU): same constructor, parameter names and state_dict layout as the
reference class, so `recognizer_best_crnn.pth` loads unchanged.  Pinned to the imported reference by tests/golden/crnn.npz."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BidirectionalLSTM(nn.Module):
    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = nn.LSTM(nIn, nHidden, bidirectional=True)
        self.embedding = nn.Linear(nHidden * 2, nOut)

    def forward(self, x):
        rec, _ = self.rnn(x)
        T, b, h = rec.shape
        return self.embedding(rec.reshape(T * b, h)).view(T, b, -1)


class CRNN(nn.Module):
    def __init__(self, imgH=32, nc=1, nclass=37, nh=256, n_rnn=2, leakyRelu=False):
        super().__init__()
        assert imgH % 16 == 0, 'imgH has to be a multiple of 16'
        ks, ps, nm = [3, 3, 3, 3, 3, 3, 2], [1, 1, 1, 1, 1, 1, 0], [64, 128, 256, 256, 512, 512, 512]
        cnn = nn.Sequential()

        def conv_relu(i, bn=False):
            cnn.add_module('conv%d' % i, nn.Conv2d(nc if i == 0 else nm[i - 1], nm[i], ks[i], 1, ps[i]))
            if bn:
                cnn.add_module('batchnorm%d' % i, nn.BatchNorm2d(nm[i]))
            cnn.add_module('relu%d' % i, nn.LeakyReLU(0.2, inplace=True) if leakyRelu else nn.ReLU(True))

        conv_relu(0)
        cnn.add_module('pooling0', nn.MaxPool2d(2, 2))
        conv_relu(1)
        cnn.add_module('pooling1', nn.MaxPool2d(2, 2))
        conv_relu(2, True)
        conv_relu(3)
        cnn.add_module('pooling2', nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
        conv_relu(4, True)
        conv_relu(5)
        cnn.add_module('pooling3', nn.MaxPool2d((2, 2), (2, 1), (0, 1)))
        conv_relu(6, True)
        self.cnn = cnn
        self.rnn = nn.Sequential(BidirectionalLSTM(512, nh, nh), BidirectionalLSTM(nh, nh, nclass))

    def forward(self, x):
        conv = self.cnn(x)
        assert conv.shape[2] == 1, "the height of conv must be 1"
        return self.rnn(conv.squeeze(2).permute(2, 0, 1))          # (T = 26, B, nclass)

    @staticmethod
    def parse_crnn_data(imgs):
        """base.py:419-425: bicubic resize to 32x100, ITU-R 601 luma."""
        x = F.interpolate(imgs, (32, 100), mode='bicubic')
        return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]

    @torch.no_grad()
    def label_vecs(self, images_lr3):
        """super_resolution.py:165-169: softmax over the classes, (T, B, 37) -> (B, 37, 1, T)."""
        logits = self(self.parse_crnn_data(images_lr3.float()))
        return torch.softmax(logits, -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2).contiguous()


def load_crnn(path, device):
    """CRNN_init (base.py:411-417): CRNN(32, 1, 37, 256) + the checkpoint's plain state dict; frozen, eval mode."""
    import os
    if not path or not os.path.isfile(path):
        raise FileNotFoundError("dpmn_amd: --arch tatt on real data needs the frozen CRNN that feeds TATT's label_vecs: "
                                "<resume>/recognizer_best_crnn.pth (super_resolution.py:92) is missing (%r)" % (path,))
    m = CRNN(32, 1, 37, 256).to(device)
    print('loading pretrained crnn model from %s' % path)
    m.load_state_dict(torch.load(path, map_location=device))
    for p in m.parameters():
        p.requires_grad = False
    return m.eval()
