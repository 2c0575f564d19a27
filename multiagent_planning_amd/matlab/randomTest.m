function [po,pf] = randomTest(N,pmin,pmax,rmin,E1,order)
% Shadows dmpc/matlab/randomTest.m (same signature): rejection-sampled start and goal sets, generated on the GPU.  MATLAB's
% global rand stream cannot be reproduced; the draws come from the library's counter-based stream, seeded from rand here.
assert(order == 2, 'only order = 2 is supported');
prm = dmpc_params_struct(0, 0.2, 15, max(rmin,0.01), pmin, pmax, 1, 1000, 100, E1, order, -5e4);   % context only
[a,b] = dmpc_mex('random_test', prm, N, pmin(:)', pmax(:)', rmin, 1/E1(3,3), floor(rand*2^52));
po = reshape(a,1,3,N); pf = reshape(b,1,3,N);
end
