"""utils.visualize — colour overlays and the IoU table of the reference's eval.py
(reference furnace/utils/visualize.py: set_img_color, show_prediction, show_img, get_colors, get_ade_colors, print_iou)."""
import numpy as np


def set_img_color(colors, background, img, gt, show255=False):
    """Paint every labelled class except `background` (and class 0) over `img`, in place."""
    for i in range(1, len(colors)):
        if i != background:
            img[gt == i] = colors[i]
    if show255:
        img[gt == 255] = 255
    return img


def show_prediction(colors, background, img, pred):
    return np.array(set_img_color(colors, background, np.array(img, np.uint8), pred))


def show_img(colors, background, img, clean, gt, *pds):
    """[clean overlay | prediction overlays ... | ground truth overlay], separated by 15-pixel black bars."""
    bar = np.zeros((np.asarray(img).shape[0], 15, 3), dtype=np.uint8)
    panels = [set_img_color(colors, background, np.array(img, np.uint8), clean)]
    panels += [set_img_color(colors, background, np.array(img, np.uint8), pd) for pd in pds]
    panels.append(set_img_color(colors, background, np.array(img, np.uint8), gt, True))
    out = panels[0]
    for p in panels[1:]:
        out = np.column_stack((out, bar, p))
    return out


def get_colors(class_num):
    return [(np.random.random((1, 3)) * 255).tolist()[0] for _ in range(class_num)]


def get_ade_colors():
    import scipy.io as sio
    colors = np.array(sio.loadmat('./color150.mat')['colors'][:, ::-1]).astype(int).tolist()
    return [[0, 0, 0]] + colors


def print_iou(iu, mean_pixel_acc, class_names=None, show_no_back=False, no_print=False):
    lines = []
    for i in range(iu.size):
        name = 'Class %d:' % (i + 1) if class_names is None else '%d %s' % (i + 1, class_names[i])
        lines.append('%-8s\t%.3f%%' % (name, iu[i] * 100))
    mean_iu, mean_iu_nb = np.nanmean(iu), np.nanmean(iu[1:])
    rule = '----------------------------     '
    if show_no_back:
        lines.append(rule + '%-8s\t%.3f%%\t%-8s\t%.3f%%\t%-8s\t%.3f%%' % (
            'mean_IU', mean_iu * 100, 'mean_IU_no_back', mean_iu_nb * 100, 'mean_pixel_ACC', mean_pixel_acc * 100))
    else:
        print(mean_pixel_acc)
        lines.append(rule + '%-8s\t%.3f%%\t%-8s\t%.3f%%' % ('mean_IU', mean_iu * 100, 'mean_pixel_ACC', mean_pixel_acc * 100))
    text = "\n".join(lines)
    if not no_print:
        print(text)
    return text
