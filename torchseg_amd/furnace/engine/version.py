__version__ = '0.1.1+mi355x'
