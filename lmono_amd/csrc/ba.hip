// placeholder: BA kernels
