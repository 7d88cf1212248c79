from .ade import ADE
