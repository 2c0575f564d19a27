function pass = ReachedGoal(p,pf,length_t,error_tol,N)
% Shadows dmpc/matlab/ReachedGoal.m (same signature): the largest goal distance at time index length_t is below error_tol (GPU reduction).
if (N > 1)
    cur = squeeze(p(:,length_t,:)); goal = squeeze(pf);
else
    cur = p(:,length_t); goal = pf(:);
end
prm = dmpc_params_struct(0, 0.2, 15, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);   % context only
pass = dmpc_mex('reached_goal', prm, reshape(cur,3,[]), reshape(goal,3,[]), error_tol);
end
